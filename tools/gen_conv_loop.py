#!/usr/bin/env python3
"""Generate gauspcc_amd/csrc/conv_loop_gfx950.inc: the hand-scheduled tile loop of k_sparse_conv (gfx950 ISA, used as
one inline-asm block).  The schedule is written down here once, with symbolic register names, so that the unrolled
copies of a step (register rings) cannot drift apart.

What shapes the schedule (tools/ubench/mfma_shadow.hip, measured on MI355X): v_mfma_f32_16x16x4_f32 runs on the SIMD's
fp32 lanes -- a VALU instruction does NOT execute in the shadow of an fp32 MFMA, it costs its 4 cycles on top of the
MFMA's 32, plus ~10 cycles every time the instruction stream switches from MFMAs to VALU work and back, from the same
wave or another one on the SIMD.  Scalar instructions, LDS and global-memory instructions do overlap.  So the step is
built to need as few VALU instructions as possible, all of them in ONE burst:

  * the product is computed transposed, D^T = W^T X^T: the weights are the MFMA A operand (pre-swizzled fragments,
    conv_weight_fragments_t), the 16 gathered neighbour rows the B operand.  Lane (g = lane/16, e = lane%16) then holds
    four physically consecutive output channels of tile row e in each accumulator quad: the running sums of a tile are
    two 16-byte LDS reads and writes per lane and four v_pk_add_f32 (was 4 + 4 two-dword accesses, 8 v_add_f32 and
    8 address computations for the four rows a lane used to touch);
  * one row byte per lane (tr holds the LDS slot = row + 1, slot 0 is the dummy row); slot address and "tile exists"
    in one v_mad_u32_u24 with a scalar multiplier of 128 or 0;
  * the ring positions of a step's header reads are immediate offsets from three pointers that move once per four
    steps (the ring carries four mirror slots behind slot 31 for that);
  * loop bookkeeping and the header staging control flow are scalar.
Per step: 16 MFMAs from a ZERO accumulator + 7 VALU (+3 per four steps); the rest is LDS / global-memory issue.

One wave owns R rows x all 32 output channels.  Per step (one tile = 16 (output row, neighbour row) pairs of one kernel
offset):
    weight fragment (W) of the tile DW = 3 steps ahead, gathered rows (X) of the tile DX steps ahead (3; CONV_ASM_DX=7 was
        built and measured in round 2: 71.6 vs 71.9 TFLOP/s on the encoder's convolutions -- the "step-start wait" the
        in-loop s_memtime stamps show is the stamp's own SMEM latency under lgkmcnt(0), not a late load); vmcnt retires in
        order, so a step waits with vmcnt(6 (DW-1)): the loads of the last DW-1 steps stay in flight, this tile's W (issued
        DW steps ago) and X (issued no later) have landed
    header words from the wave's LDS ring: neighbour rows + offset of tile u+D+1, output slot of tile u+1
    the PREVIOUS tile's products are added to its rows' running sums (read from LDS during the previous step) and
        written back; then this tile's rows are read -- LDS operations of a wave execute in program order, so
        consecutive tiles may share output rows
    header staging: three wide loads every 16 tiles, moved into the ring 8 tiles later
Registers v16.. are fixed here (declared as clobbers), the rest is the compiler's.  Register tuples start on even
registers (gfx90a+ requirement).
"""
import os

EXP = int(os.environ.get("CONV_ASM_EXP", "0"))   # developer experiments: 1 no sums in LDS, 2 no weight loads, 4 no gathers,
                                                 # 16 every odd tile multiplies with the previous tile's weights and loads none (timing of "two tiles share a fragment", WRONG results)
                                                 # 32 as 16, but the odd tiles still issue four loads -- of one 16-byte address for all lanes, into scratch registers (issue slots kept, L1 traffic gone)
DX = int(os.environ.get("CONV_ASM_DX", "3"))     # issue distance of the gathered rows (steps); X ring = DX + 1 sets (7: measured, no gain)
DW = 3                                           # issue distance of the weight fragments; W ring = DW + 1 sets
HALF = 0   # (the half-channel loop of round 4 -- 16 of the 32 output channels per wave, measured -6 .. -35 % -- left the tree with the kernel's freeze in round 6; the HALF branches below document what it changed)
                                                     # 16-byte piece of running sums per lane, half a weight fragment (two loads); two such waves share a block's tile list
ROWB = int(os.environ.get("CONV_ASM_ROWB", "64" if HALF else "144"))   # LDS bytes per row of running sums (128 + one 16-byte pad: bank rotation, network.hip; half rows: 64, no pad)
assert DX in (3, 7) and ROWB % 16 == 0 and ROWB >= (64 if HALF else 128)
NX, NW = DX + 1, DW + 1
_x0 = 16
_w0 = _x0 + 8 * NX
_c0 = _w0 + 16 * NW
_m0 = _c0 + 32                         # first register behind X | W | C (2 x 8) | S (2 x 8)

V = dict(
    X=tuple(_x0 + 8 * i for i in range(NX)),     # gathered rows: NX sets x 8 regs
    W=tuple(_w0 + 16 * i for i in range(NW)),    # weight fragments: NW sets x 16 regs
    C=(_c0, _c0 + 8),                  # products: c0 = C..C+3, c1 = C+4..C+7
    S=(_c0 + 16, _c0 + 24),            # running sums of the tile's rows: 2 sets x 8 regs
    jn=_m0, on=_m0 + 1, rb=_m0 + 2, ao=_m0 + 3, bo=_m0 + 4,
    ra=(_m0 + 5, _m0 + 6),             # LDS address of this lane's 16 bytes in the tile row's slot
    t0=_m0 + 7,
    stj=_m0 + 8, str=_m0 + 12, sto=_m0 + 13,     # staged header batch (stj: 4 registers, even-aligned)
    accb=_m0 + 14, hjb=_m0 + 15, hrb=_m0 + 16, hob=_m0 + 17, goff=_m0 + 18, loff=_m0 + 19,
    hjp=_m0 + 20, hrp=_m0 + 21, hop=_m0 + 22,
    sgr=_m0 + 23, sgo=_m0 + 24, swj=_m0 + 25, swr=_m0 + 26, swo=_m0 + 27,
    zero=_m0 + 28, dummy=_m0 + 30,     # EXP 32: a zero offset and 16 scratch registers (even-aligned)
)
CLOBBER_V = list(range(16, _m0 + 28 + (20 if EXP & 32 else 0)))
NSTEP = NX                             # a multiple of NX, NW and 2
LPS = 4 if HALF else 6                 # loads a step issues: 2 gathers + the (half) weight fragment
WINDOW = LPS * (DW - 1)                # tile loads that may stay in flight across a step start
# Header ring: 32 slots = two batches of 16 tiles, followed by a mirror of slots 0..15 so that the NSTEP steps of an
# iteration reach every slot they read by immediate offsets from pointers that move once per iteration.
# Byte offsets inside the ring (must match network.hip: HDR_R, HDR_O):
RING_SLOTS = 48
RING_R = RING_SLOTS * 64               # neighbour rows: 64 B a slot
RING_O = RING_R + RING_SLOTS * 16      # output slots: 16 B a slot; then the kernel offsets: 4 B a slot


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mfma(c, w, x, first):
    return f"v_mfma_f32_16x16x4_f32 {vr(c, 4)}, v{w}, v{x}, {'0' if first else vr(c, 4)}"


def loads_x(xset):
    X = V["X"][xset]
    o = [f"global_load_dwordx4 {vr(X, 4)}, v{V['ao']}, %[in]",
         f"global_load_dwordx4 {vr(X + 4, 4)}, v{V['ao']}, %[in] offset:64"]
    return ["s_nop 0"] * 2 if EXP & 4 else o


def loads_w(wset):
    W = V["W"][wset]
    o = [f"global_load_dwordx4 {vr(W + 4 * i, 4)}, v{V['bo']}, %[w]" + (f" offset:{1024 * i}" if i else "") for i in range(2 if HALF else 4)]
    return ["s_nop 0"] * len(o) if EXP & 2 else o


def loads_w_dummy():
    D = V["dummy"]
    return [f"global_load_dwordx4 {vr(D + 4 * i, 4)}, v{V['zero']}, %[w]" for i in range(4)]


def addr_x():
    return [f"v_lshl_add_u32 v{V['ao']}, v{V['jn']}, 7, v{V['goff']}"]


def addr_w():
    return [f"v_lshl_add_u32 v{V['bo']}, v{V['on']}, 12, v{V['loff']}"]


def sum_adds(sset, cp):
    if EXP & 1:
        return []
    S = V["S"][sset]
    if os.environ.get("CONV_ASM_PK", "1") == "0":      # experiment: eight scalar adds instead of four packed ones
        return [f"v_add_f32 v{S + i}, v{S + i}, v{cp + i}" for i in range(8)]
    return [f"v_pk_add_f32 {vr(S + 2 * i, 2)}, {vr(S + 2 * i, 2)}, {vr(cp + 2 * i, 2)}" for i in range(2 if HALF else 4)]


def sum_writes(sset):
    if EXP & 1:
        return []
    S, ra = V["S"][sset], V["ra"][sset]
    return [f"ds_write_b128 v{ra}, {vr(S, 4)}"] + ([] if HALF else [f"ds_write_b128 v{ra}, {vr(S + 4, 4)} offset:64"])


def sum_reads(sset):
    if EXP & 1:
        return []
    S, ra = V["S"][sset], V["ra"][sset]
    return [f"ds_read_b128 {vr(S, 4)}, v{ra}"] + ([] if HALF else [f"ds_read_b128 {vr(S + 4, 4)}, v{ra} offset:64"])


def header_reads(du):
    # neighbour rows of tile u+du+DX+1 and offset of tile u+du+DW+1 (their loads are issued in the next step), output slot of tile u+du+1
    assert 24 + DX + 1 + du < RING_SLOTS
    return [f"ds_read_b32 v{V['jn']}, v{V['hjp']} offset:{64 * (DX + 1 + du)}",
            f"ds_read_b32 v{V['on']}, v{V['hop']} offset:{4 * (DW + 1 + du)}",
            f"ds_read_u8 v{V['rb']}, v{V['hrp']} offset:{16 * (1 + du)}"]


def ring_pointers():
    # ring position of tile u (u % NSTEP == 0): the steps of an iteration reach slots su+1 .. su+DX+NSTEP < 48 by immediate offsets
    return ["s_and_b32 %[t0], %[u], 31",
            f"v_lshl_add_u32 v{V['hjp']}, %[t0], 6, v{V['hjb']}",
            f"v_lshl_add_u32 v{V['hrp']}, %[t0], 4, v{V['hrb']}",
            f"v_lshl_add_u32 v{V['hop']}, %[t0], 2, v{V['hob']}"]


def staging_fetch(label):
    # header batch (u / 16) + 2 is fetched at the middle step of batch u / 16 ...
    return [
        "s_and_b32 %[t0], %[u], 15",
        "s_cmp_eq_u32 %[t0], 8",
        f"s_cbranch_scc0 {label}_nofetch%=",
        "s_lshr_b32 %[t1], %[u], 4",
        "s_add_u32 %[t1], %[t1], 2",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 10, v{V['loff']}",
        f"global_load_dwordx4 {vr(V['stj'], 4)}, v{V['t0']}, %[tj]",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 8, v{V['sgr']}",
        f"global_load_dword v{V['str']}, v{V['t0']}, %[tr]",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 6, v{V['sgo']}",
        f"global_load_dword v{V['sto']}, v{V['t0']}, %[toc]",
        f"{label}_nofetch%=:",
    ]


def staging_store(label, younger_loads):
    # ... and moved into the ring eight steps later, at the first step of batch (u / 16): the half of the ring it replaces
    # (batch (u / 16) - 1) was last read before this step, and the steps that read its first slots are >= 8 steps away.
    # A batch that lands in slots 0..15 is also written to their mirror behind slot 31.
    return [
        "s_and_b32 %[t0], %[u], 15",
        "s_cmp_eq_u32 %[t0], 0",
        f"s_cbranch_scc0 {label}_nostore%=",
        "s_cmp_eq_u32 %[u], 0",
        f"s_cbranch_scc1 {label}_nostore%=",
        f"s_waitcnt vmcnt({younger_loads})",          # the staged loads are eight steps old
        "s_lshr_b32 %[t1], %[u], 4",
        "s_add_u32 %[t1], %[t1], 1",
        "s_and_b32 %[t1], %[t1], 1",
        f"v_and_b32 v{V['sto']}, 0xffff, v{V['sto']}",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 10, v{V['swj']}",
        f"ds_write_b128 v{V['t0']}, {vr(V['stj'], 4)}",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 8, v{V['swr']}",
        f"ds_write_b32 v{V['t0']}, v{V['str']}",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 6, v{V['swo']}",
        f"ds_write_b32 v{V['t0']}, v{V['sto']}",
        "s_cmp_eq_u32 %[t1], 0",
        f"s_cbranch_scc0 {label}_nostore%=",
        f"ds_write_b128 v{V['swj']}, {vr(V['stj'], 4)} offset:2048",
        f"ds_write_b32 v{V['swr']}, v{V['str']} offset:512",
        f"ds_write_b32 v{V['swo']}, v{V['sto']} offset:128",
        f"{label}_nostore%=:",
    ]


def step_wait(du, label):
    """The loads of the last DW - 1 steps may stay in flight; everything older (this tile's W, and its X from long before)
    must have landed.  For DW steps after a header fetch (issued behind the tile loads of step 0) its 3 loads are inside
    that window as well.  Every LDS operation of the previous step was issued behind its first MFMA pair: long done."""
    if EXP & 16 and not EXP & 32:
        # loads of the last DW - 1 = 2 steps: one step with W (6) and one without (2)
        win = 8
        if not 1 <= du <= DW:
            return [f"s_waitcnt vmcnt({win}) lgkmcnt(0)"]
        return ["s_and_b32 %[t0], %[u], 15", "s_cmp_eq_u32 %[t0], 8", f"s_cbranch_scc0 {label}_wa%=", f"s_waitcnt vmcnt({win + 3}) lgkmcnt(0)", f"s_branch {label}_wb%=",
                f"{label}_wa%=:", f"s_waitcnt vmcnt({win}) lgkmcnt(0)", f"{label}_wb%=:"]
    if not 1 <= du <= DW:
        return [f"s_waitcnt vmcnt({WINDOW}) lgkmcnt(0)"]
    return ["s_and_b32 %[t0], %[u], 15",
            "s_cmp_eq_u32 %[t0], 8",
            f"s_cbranch_scc0 {label}_wa%=",
            f"s_waitcnt vmcnt({WINDOW + 3}) lgkmcnt(0)",
            f"s_branch {label}_wb%=",
            f"{label}_wa%=:",
            f"s_waitcnt vmcnt({WINDOW}) lgkmcnt(0)",
            f"{label}_wb%=:"]


STAMPS = ("sa", "sb", "sc", "sd")      # EXP & 8: s_memtime at four points of every step (developer timing build)


STAMP_S = {f"{n}{p}": 60 + 2 * (4 * p + i) for p in (0, 1) for i, n in enumerate(STAMPS)}   # fixed SGPR pairs s[60:75]


def stamp(name, par):
    r = STAMP_S[f"{name}{par}"]
    return [f"s_memtime s[{r}:{r + 1}]"] if EXP & 8 else []


def stamp_collect(par):
    """Differences of the previous step's stamps (all landed: the step start waited for lgkmcnt(0)) into the accumulators:
    wait = sb - sa, burst = sc - sb, loads = sd - sc, rest = this step's sa - sd."""
    if not EXP & 8:
        return []
    q = 1 - par
    o = []
    for acc, (hi, lo) in (("w0", ("sb", "sa")), ("w1", ("sc", "sb")), ("w2", ("sd", "sc"))):
        o += [f"s_sub_u32 %[t0], s{STAMP_S[hi + str(q)]}, s{STAMP_S[lo + str(q)]}", f"s_add_u32 %[{acc}], %[{acc}], %[t0]"]
    o += [f"s_sub_u32 %[t0], s{STAMP_S['sa' + str(par)]}, s{STAMP_S['sd' + str(q)]}", "s_add_u32 %[w3], %[w3], %[t0]"]
    return o


def step(du):
    cur, prv = du % 2, 1 - du % 2
    X, W, CC, CP = V["X"][du % NX], V["W"][(du - (du & 1 if EXP & 48 else 0)) % NW], V["C"][cur], V["C"][prv]
    label = f"s{du}"

    def mf(kk, first=False):
        return [mfma(CC, W + kk, X + kk, first)] + ([] if HALF else [mfma(CC + 4, W + 8 + kk, X + kk, first)])

    nsx, nsw = (du + DX) % NX, (du + DW) % NW    # the sets consumed by the previous step receive tiles u+du+DX / u+du+DW
    o = [f"; ---- step: tile u+{du}: X set {du % NX}, W set {du % NW}, C/S set {cur}"] + stamp("sa", cur) + step_wait(du, label) + stamp("sb", cur) + stamp_collect(cur)
    # slot multiplier of this tile: the row pitch if it exists, 0 (the dummy slot) past the end of the block's list
    o += [f"s_add_u32 %[t0], %[u], {du}", "s_cmp_lt_u32 %[t0], %[nt]", f"s_cselect_b32 %[t1], {ROWB}, 0"]
    o += mf(0, True)
    if HALF:
        # one accumulator chain: the previous step's LAST MFMA (which wrote the products the burst below adds) was issued one MFMA ago --
        # a second MFMA of the new chain in front of the burst keeps the 11 wait states an 8-pass MFMA result needs before a VALU read
        o += mf(1)
    # the step's only VALU burst: previous tile's products onto its rows' sums, addresses of the loads and of this tile's slot
    o += (ring_pointers() if du == 0 else [])
    o += sum_adds(prv, CP) + addr_x() + addr_w()
    o += [f"v_mad_u32_u24 v{V['ra'][cur]}, v{V['rb']}, %[t1], v{V['accb']}"]
    # memory instructions issue slowly (measured: ~20 cycles an LDS, ~13 a global load instruction, during which an in-order
    # wave issues nothing else) but, unlike VALU work, they do overlap a running MFMA: one in front of each remaining MFMA
    mem = sum_writes(prv) + sum_reads(cur) + header_reads(du) + loads_x(nsx) + ((loads_w_dummy() if EXP & 32 else []) if (EXP & 48) and (du + DW) % 2 == 1 else loads_w(nsw))
    rest = [m for kk in range(2 if HALF else 1, 8) for m in mf(kk)]
    spread = os.environ.get("CONV_ASM_SPREAD", "1") != "0"
    if not spread:
        o += mem[:-6] + stamp("sc", cur) + rest[:2] + mem[-6:] + (staging_fetch(label) if du == 0 else []) + stamp("sd", cur)
        o += rest[2:10] + (staging_store(label, WINDOW + LPS) if du == 0 else []) + rest[10:]
        return o
    stag = (staging_fetch(label) + staging_store(label, (8 if EXP & 16 and not EXP & 32 else WINDOW) + (2 if EXP & 16 and not EXP & 32 else LPS)) if du == 0 else [])
    for i, m in enumerate(rest):
        if i < len(mem):
            o.append(mem[i])
        if i == len(mem):
            o += stag
        if i == 7:
            o += stamp("sc", cur)
        o.append(m)
    if len(mem) >= len(rest):    # (the half-channel loop has fewer MFMAs than memory instructions: the rest of them, and the staging, behind the last MFMA)
        o += mem[len(rest):] + stag
    o += stamp("sd", cur)
    return o


def build():
    L = V
    o = [
        "; ---- per-lane constants",
        f"v_and_b32 v{L['t0']}, 15, %[lane]",                       # e
        f"v_lshrrev_b32 v{L['goff']}, 4, %[lane]",                  # g
        f"v_lshlrev_b32 v{L['goff']}, 4, v{L['goff']}",             # 16 g: byte offset of this lane's channels in a row half
        f"v_add_u32 v{L['accb']}, %[acc], v{L['goff']}",
        f"v_lshl_add_u32 v{L['hjb']}, v{L['t0']}, 2, %[hdr]",
        f"v_add_u32 v{L['hrb']}, %[hdr], v{L['t0']}",
        f"v_add_u32 v{L['hrb']}, {RING_R}, v{L['hrb']}",
        f"v_mov_b32 v{L['hob']}, %[hdr]",
        f"v_add_u32 v{L['hob']}, {RING_O}, v{L['hob']}",
        f"v_lshlrev_b32 v{L['loff']}, 4, %[lane]",
        f"v_lshlrev_b32 v{L['sgr']}, 2, %[lane]",
        f"v_lshlrev_b32 v{L['sgo']}, 2, v{L['t0']}",
        f"v_add_u32 v{L['swj']}, %[hdr], v{L['loff']}",
        f"v_add_u32 v{L['swr']}, %[hdr], v{L['sgr']}",
        f"v_add_u32 v{L['swr']}, {RING_R}, v{L['swr']}",
        f"v_add_u32 v{L['swo']}, v{L['hob']}, v{L['sgo']}",
        f"v_mov_b32 v{L['ra'][1]}, v{L['accb']}",                   # "previous tile" of step 0: the dummy slot
    ] + ([f"v_mov_b32 v{L['zero']}, 0"] if EXP & 32 else []) + [
        "; ---- pipeline prologue: X of tiles 0..DX-1 and W of tiles 0..DW-1 in flight; headers j(DX), o(DW), slot(0) in registers",
        "s_waitcnt lgkmcnt(0)",
    ]
    for t in range(DX):
        o += [f"ds_read_b32 v{L['jn']}, v{L['hjb']} offset:{64 * t}"] + ([f"ds_read_b32 v{L['on']}, v{L['hob']} offset:{4 * t}"] if t < DW else []) + ["s_waitcnt lgkmcnt(0)"]
        o += addr_x() + loads_x(t) + (addr_w() + loads_w(t) if t < DW else [])
    o += [
        f"ds_read_b32 v{L['jn']}, v{L['hjb']} offset:{64 * DX}",
        f"ds_read_b32 v{L['on']}, v{L['hob']} offset:{4 * DW}",
        f"ds_read_u8 v{L['rb']}, v{L['hrb']}",
        "s_mov_b32 %[u], 0",
    ] + ([f"s_mov_b32 %[w{i}], 0" for i in range(4)] + [x for n in STAMPS for x in stamp(n, 1)] if EXP & 8 else []) + [
        "conv_loop%=:",
    ]
    for du in range(NSTEP):
        o += step(du)
        if du % 2 == 1 and du != NSTEP - 1:   # leave after an even number of steps when the list is exhausted (set 1 holds the last products either way)
            o += [f"s_add_u32 %[t0], %[u], {du + 1}", "s_cmp_ge_u32 %[t0], %[nt]", "s_cbranch_scc1 conv_drain%="]
    o += [
        f"s_add_u32 %[u], %[u], {NSTEP}",
        "s_cmp_lt_u32 %[u], %[nt]",
        "s_cbranch_scc1 conv_loop%=",
        "conv_drain%=:",
        "; ---- drain: products of the last step (set 1) onto the sums read during it",
        "s_waitcnt lgkmcnt(0)",
        "s_nop 15",                    # MFMA result -> VALU read needs 11 wait states after an 8-pass MFMA; nothing else separates them here
    ]
    o += sum_adds(1, V["C"][1]) + sum_writes(1)
    o += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return o


def main():
    path = os.environ.get("CONV_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gauspcc_amd", "csrc", "conv_looph_gfx950.inc" if HALF else "conv_loop_gfx950.inc")
    sfx = "H" if HALF else ""
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_conv_loop.py" + (" (CONV_ASM_HALF=1)" if HALF else "") + " -- do not edit.  gfx950 ISA of the k_sparse_conv tile loops.\n")
        for name in (f"CONV_LOOP{sfx}_ASM",):
            o = build()
            f.write(f"#define {name} \\\n")
            for ln in o:
                f.write('    "' + ln + '\\n" \\\n')
            f.write('    ""\n')
            print(f"{name}: {len(o)} lines")
        f.write(f"#define CONV_LOOP{sfx}_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + (", " + ", ".join(f'"s{i}"' for i in range(60, 76)) if EXP & 8 else "")
                + ', "vcc", "scc", "memory"\n')
        f.write(f"#define CONV_LDS_ROW_BYTES{'_H' if HALF else ''} {ROWB}\n")
        if EXP & 8:
            f.write("#define CONV_LOOP_STAMPS 1\n")
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
