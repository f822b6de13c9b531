#!/usr/bin/env python3
"""Generate gauspcc_amd/csrc/conv_loop2_gfx950.inc: the PAIR-STEP tile loop of k_sparse_conv (gfx950 ISA, one inline-asm
block) -- round 3's successor of tools/gen_conv_loop.py for tile lists whose runs were padded to an even number of tiles
(tiles.hip, `paired` pools: blocks taller than 64 rows).

Why pairs.  With the running sums at a 144-byte LDS pitch (no bank conflicts), the ablations of the one-tile-a-step loop
read (encoder convolutions, 1 M-point cloud, TFLOP/s): as built 73.3; every second tile re-using the previous tile's
weight fragment and issuing no weight loads 83.0 (+13 %); the same with the four loads still issued but to one address
76.5 (+4 %); no weight loads at all 88.0.  The four `global_load_dwordx4` of a fragment cost the in-order wave more as
ISSUE slots than as L1 traffic, so they have to disappear from the instruction stream, and the count of loads a step
issues has to stay a constant (`s_waitcnt vmcnt(N)` takes an immediate).  Both hold if a step is TWO tiles of one kernel
offset: 4 + 4 gather loads, ONE fragment (4 loads), 32 MFMAs, one step-start wait, one VALU burst.  The tile builder pads
every (block, offset) run to an even number of tiles; the odd run's last step finds its second tile empty (its slot bytes
are all 0: `v_readfirstlane` of the byte lane 0 read + one scalar branch) and runs a body without that tile's 16 MFMAs --
same loads (the empty tile's neighbour rows are row 0: one cache line), same waits.

Sums stay offset-ascending per row: the tiles of a step belong to one offset and hold disjoint rows, steps run in list
order, and the LDS operations of a wave execute in program order.  Bit-identical to every other conv kernel.

Per step (tiles a = u + 2 du, b = a + 1; parity p = du % 2):
    wait vmcnt(8 (DW - 1)): the loads of the last DW - 1 steps may stay in flight
    products of the previous step's two tiles are added to the sums read during it and written back; this step's rows are read
    header words from the wave's LDS ring: neighbour rows of the two tiles DX + 1 steps ahead, offset of the pair DW + 1
        steps ahead, output slots of the next step's tiles
    X of the pair DX steps ahead (4 loads), W of the pair DW steps ahead (4 loads)
Registers v16.. are fixed here (declared as clobbers).
"""
import os

DX, DW = 3, 3
EXP2 = int(os.environ.get("CONV_ASM_EXP2", "0"))   # developer ablations of the pair loop (WRONG results): 1 = conflict-free LDS slots, 2 = no running sums
                                                     # (no LDS read / add / write), 4 = no weight loads in the loop, 8 = no gathers in the loop, 16 = no header reads
NX, NW = DX + 1, DW + 1
ROWB = int(os.environ.get("CONV_ASM_ROWB", "144"))
_x0 = 16
_w0 = _x0 + 16 * NX                    # X: NX sets x (tile a: 8, tile b: 8)
_c0 = _w0 + 16 * NW                    # W: NW sets x 16
_s0 = _c0 + 32                         # C: 2 parities x (a: 8, b: 8)
_m0 = _s0 + 32                         # S: 2 parities x (a: 8, b: 8)

V = dict(
    jn=(_m0, _m0 + 1), on=_m0 + 2, rb=(_m0 + 3, _m0 + 4), ao=(_m0 + 5, _m0 + 6), bo=_m0 + 7,
    ra=((_m0 + 8, _m0 + 9), (_m0 + 10, _m0 + 11)),     # [parity][tile]
    t0=_m0 + 12,
    stj=_m0 + 14, str=_m0 + 18, sto=_m0 + 19,          # staged header batch (stj: 4 registers, even-aligned)
    accb=_m0 + 20, hjb=_m0 + 21, hrb=_m0 + 22, hob=_m0 + 23, goff=_m0 + 24, loff=_m0 + 25,
    hjp=_m0 + 26, hrp=_m0 + 27, hop=_m0 + 28,
    sgr=_m0 + 29, sgo=_m0 + 30, swj=_m0 + 31, swr=_m0 + 32, swo=_m0 + 33,
)
assert _m0 % 2 == 0 and V["stj"] % 2 == 0
CLOBBER_V = list(range(16, _m0 + 34))
assert CLOBBER_V[-1] < 256
NSTEP = NX
WINDOW = 8 * (DW - 1)
RING_SLOTS = 48
RING_R = RING_SLOTS * 64
RING_O = RING_R + RING_SLOTS * 16


def X(s, t):
    return _x0 + 16 * s + 8 * t


def W(s):
    return _w0 + 16 * s


def C(p, t):
    return _c0 + 16 * p + 8 * t


def S(p, t):
    return _s0 + 16 * p + 8 * t


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mfma(c, w, x, first):
    return f"v_mfma_f32_16x16x4_f32 {vr(c, 4)}, v{w}, v{x}, {'0' if first else vr(c, 4)}"


def mf(cc, wset, xx, kk, first=False):
    # output half 0 from fragment registers 0..7, half 1 from 8..15
    return [mfma(cc, W(wset) + kk, xx + kk, first), mfma(cc + 4, W(wset) + 8 + kk, xx + kk, first)]


def loads_x(xset, t):
    b = X(xset, t)
    return [f"global_load_dwordx4 {vr(b, 4)}, v{V['ao'][t]}, %[in]", f"global_load_dwordx4 {vr(b + 4, 4)}, v{V['ao'][t]}, %[in] offset:64"]


def loads_w(wset):
    b = W(wset)
    return [f"global_load_dwordx4 {vr(b + 4 * i, 4)}, v{V['bo']}, %[w]" + (f" offset:{1024 * i}" if i else "") for i in range(4)]


def addr_x(t):
    return [f"v_lshl_add_u32 v{V['ao'][t]}, v{V['jn'][t]}, 7, v{V['goff']}"]


def addr_w():
    return [f"v_lshl_add_u32 v{V['bo']}, v{V['on']}, 12, v{V['loff']}"]


def sum_adds(p, t):
    s, c = S(p, t), C(p, t)
    return [f"v_pk_add_f32 {vr(s + 2 * i, 2)}, {vr(s + 2 * i, 2)}, {vr(c + 2 * i, 2)}" for i in range(4)]


def sum_writes(p, t):
    s, ra = S(p, t), V["ra"][p][t]
    return [f"ds_write_b128 v{ra}, {vr(s, 4)}", f"ds_write_b128 v{ra}, {vr(s + 4, 4)} offset:64"]


def sum_reads(p, t):
    s, ra = S(p, t), V["ra"][p][t]
    return [f"ds_read_b128 {vr(s, 4)}, v{ra}", f"ds_read_b128 {vr(s + 4, 4)}, v{ra} offset:64"]


def header_reads(du):
    # neighbour rows of the pair DX + 1 steps ahead and its offset DW + 1 steps ahead (their loads are issued in the next
    # step), output slots of the next step's pair.  Slots relative to the ring position of tile u.
    assert 24 + 2 * (DX + 1 + du) + 1 < RING_SLOTS
    ja, oa, ra = 2 * (DX + 1 + du), 2 * (DW + 1 + du), 2 * (1 + du)
    return [f"ds_read_b32 v{V['jn'][0]}, v{V['hjp']} offset:{64 * ja}",
            f"ds_read_b32 v{V['jn'][1]}, v{V['hjp']} offset:{64 * (ja + 1)}",
            f"ds_read_b32 v{V['on']}, v{V['hop']} offset:{4 * oa}",
            f"ds_read_u8 v{V['rb'][0]}, v{V['hrp']} offset:{16 * ra}",
            f"ds_read_u8 v{V['rb'][1]}, v{V['hrp']} offset:{16 * (ra + 1)}"]


def ring_pointers():
    return ["s_and_b32 %[t0], %[u], 31",
            f"v_lshl_add_u32 v{V['hjp']}, %[t0], 6, v{V['hjb']}",
            f"v_lshl_add_u32 v{V['hrp']}, %[t0], 4, v{V['hrb']}",
            f"v_lshl_add_u32 v{V['hop']}, %[t0], 2, v{V['hob']}"]


def staging_fetch(label):
    # header batch (u / 16) + 2 is fetched at tile 8 of batch u / 16 ...
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
    # ... and moved into the ring eight tiles later, at the first tile of batch u / 16: the half of the ring it replaces was
    # last read before this step.  A batch that lands in slots 0..15 is also written to their mirror behind slot 31.
    return [
        "s_and_b32 %[t0], %[u], 15",
        "s_cmp_eq_u32 %[t0], 0",
        f"s_cbranch_scc0 {label}_nostore%=",
        "s_cmp_eq_u32 %[u], 0",
        f"s_cbranch_scc1 {label}_nostore%=",
        f"s_waitcnt vmcnt({younger_loads})",
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
    """The loads of the last DW - 1 steps (8 each) may stay in flight; for the DW steps behind a header fetch (issued behind
    the loads of step 0 of the iteration with u % 16 == 8) its three loads are younger than what the step needs as well."""
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


def interleave(mfmas, mem, extra_at_end_of_mem):
    """one memory instruction in front of each MFMA; `extra_at_end_of_mem` (the header staging) behind the last of them"""
    o = []
    for i, m in enumerate(mfmas):
        if i < len(mem):
            o.append(mem[i])
        if i == len(mem):
            o += extra_at_end_of_mem
        o.append(m)
    if len(mem) >= len(mfmas):
        o += mem[len(mfmas):] + extra_at_end_of_mem
    return o


def step(du):
    p, q = du % 2, 1 - du % 2
    xs, ws = du % NX, du % NW
    nsx, nsw = (du + DX) % NX, (du + DW) % NW
    label = f"s{du}"
    o = [f"; ---- step: tiles u+{2 * du}, u+{2 * du + 1}: X set {xs}, W set {ws}, C/S parity {p}"] + step_wait(du, label)
    # slot multiplier of the pair: the row pitch if it exists, 0 (the dummy slot) past the end of the block's list (nt is even)
    o += [f"s_add_u32 %[t0], %[u], {2 * du}", "s_cmp_lt_u32 %[t0], %[nt]", f"s_cselect_b32 %[t1], {ROWB}, 0"]
    # is the second tile of the pair an empty one (the padding of an odd run)?  its slot bytes are all 0
    o += [f"v_readfirstlane_b32 %[t0], v{V['rb'][1]}", "s_cmp_eq_u32 %[t0], 0", f"s_cbranch_scc1 {label}_half%="]

    def body(full):
        b = []
        b += mf(C(p, 0), ws, X(xs, 0), 0, True)
        if full:
            b += mf(C(p, 1), ws, X(xs, 1), 0, True)
        # the step's only VALU burst
        b += (ring_pointers() if du == 0 else [])
        b += ([] if EXP2 & 2 else sum_adds(q, 0) + sum_adds(q, 1)) + addr_x(0) + addr_x(1) + addr_w()
        if EXP2 & 1:   # ablation (WRONG results): every tile's rows are LDS slots 0..15 -- what the bank conflicts of gathered slots cost
            b += [f"v_mad_u32_u24 v{V['ra'][p][0]}, v{V['sgo']}, {ROWB // 4}, v{V['accb']}",
                  f"v_mad_u32_u24 v{V['ra'][p][1]}, v{V['sgo']}, {ROWB // 4}, v{V['accb']}"]
        else:
            b += [f"v_mad_u32_u24 v{V['ra'][p][0]}, v{V['rb'][0]}, %[t1], v{V['accb']}",
                  f"v_mad_u32_u24 v{V['ra'][p][1]}, v{V['rb'][1]}, %[t1], v{V['accb']}"]
        nop = lambda lst: ["s_nop 0"] * len(lst)
        sw, sr = sum_writes(q, 0) + sum_writes(q, 1), sum_reads(p, 0) + sum_reads(p, 1)
        hr, lx, lw = header_reads(du), loads_x(nsx, 0) + loads_x(nsx, 1), loads_w(nsw)
        mem = ((nop(sw) + nop(sr) if EXP2 & 2 else sw + sr) + (nop(hr) if EXP2 & 16 else hr) + (nop(lx) if EXP2 & 8 else lx) + (nop(lw) if EXP2 & 4 else lw))
        rest = []
        for kk in range(1, 8):
            rest += mf(C(p, 0), ws, X(xs, 0), kk)
            if full:
                rest += mf(C(p, 1), ws, X(xs, 1), kk)
        tag = f"{label}{'f' if full else 'h'}"
        stag = (staging_fetch(tag) + staging_store(tag, WINDOW + 8)) if du == 0 else []
        b += interleave(rest, mem, stag)
        return b

    o += body(True)
    o += [f"s_branch {label}_join%=", f"{label}_half%=:"]
    o += body(False)
    o += [f"{label}_join%=:"]
    return o


def build():
    o = [
        "; ---- per-lane constants",
        f"v_and_b32 v{V['t0']}, 15, %[lane]",                       # e
        f"v_lshrrev_b32 v{V['goff']}, 4, %[lane]",                  # g
        f"v_lshlrev_b32 v{V['goff']}, 4, v{V['goff']}",             # 16 g: byte offset of this lane's channels in a row half
        f"v_add_u32 v{V['accb']}, %[acc], v{V['goff']}",
        f"v_lshl_add_u32 v{V['hjb']}, v{V['t0']}, 2, %[hdr]",
        f"v_add_u32 v{V['hrb']}, %[hdr], v{V['t0']}",
        f"v_add_u32 v{V['hrb']}, {RING_R}, v{V['hrb']}",
        f"v_mov_b32 v{V['hob']}, %[hdr]",
        f"v_add_u32 v{V['hob']}, {RING_O}, v{V['hob']}",
        f"v_lshlrev_b32 v{V['loff']}, 4, %[lane]",
        f"v_lshlrev_b32 v{V['sgr']}, 2, %[lane]",
        f"v_lshlrev_b32 v{V['sgo']}, 2, v{V['t0']}",
        f"v_add_u32 v{V['swj']}, %[hdr], v{V['loff']}",
        f"v_add_u32 v{V['swr']}, %[hdr], v{V['sgr']}",
        f"v_add_u32 v{V['swr']}, {RING_R}, v{V['swr']}",
        f"v_add_u32 v{V['swo']}, v{V['hob']}, v{V['sgo']}",
        f"v_mov_b32 v{V['ra'][1][0]}, v{V['accb']}",                # "previous pair" of step 0: the dummy slot
        f"v_mov_b32 v{V['ra'][1][1]}, v{V['accb']}",
        "; ---- pipeline prologue: X of pairs 0..DX-1 and W of pairs 0..DW-1 in flight; headers j(pair DX), o(pair DW), slots(pair 0) in registers",
        "s_waitcnt lgkmcnt(0)",
    ]
    for s in range(DX):
        o += [f"ds_read_b32 v{V['jn'][0]}, v{V['hjb']} offset:{64 * 2 * s}", f"ds_read_b32 v{V['jn'][1]}, v{V['hjb']} offset:{64 * (2 * s + 1)}"]
        o += [f"ds_read_b32 v{V['on']}, v{V['hob']} offset:{4 * 2 * s}"] if s < DW else []
        o += ["s_waitcnt lgkmcnt(0)"]
        o += addr_x(0) + addr_x(1) + loads_x(s, 0) + loads_x(s, 1) + (addr_w() + loads_w(s) if s < DW else [])
    o += [
        f"ds_read_b32 v{V['jn'][0]}, v{V['hjb']} offset:{64 * 2 * DX}",
        f"ds_read_b32 v{V['jn'][1]}, v{V['hjb']} offset:{64 * (2 * DX + 1)}",
        f"ds_read_b32 v{V['on']}, v{V['hob']} offset:{4 * 2 * DW}",
        f"ds_read_u8 v{V['rb'][0]}, v{V['hrb']}",
        f"ds_read_u8 v{V['rb'][1]}, v{V['hrb']} offset:16",
        "s_mov_b32 %[u], 0",
        "conv_loop%=:",
    ]
    for du in range(NSTEP):
        o += step(du)
        if du % 2 == 1 and du != NSTEP - 1:   # leave after an even number of steps when the list is exhausted (parity 1 holds the last products either way)
            o += [f"s_add_u32 %[t0], %[u], {2 * (du + 1)}", "s_cmp_ge_u32 %[t0], %[nt]", "s_cbranch_scc1 conv_drain%="]
    o += [
        f"s_add_u32 %[u], %[u], {2 * NSTEP}",
        "s_cmp_lt_u32 %[u], %[nt]",
        "s_cbranch_scc1 conv_loop%=",
        "conv_drain%=:",
        "; ---- drain: products of the last step (parity 1) onto the sums read during it",
        "s_waitcnt lgkmcnt(0)",
        "s_nop 15",                    # MFMA result -> VALU read needs 11 wait states after an 8-pass MFMA; nothing else separates them here
    ]
    o += sum_adds(1, 0) + sum_adds(1, 1) + sum_writes(1, 0) + sum_writes(1, 1)
    o += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return o


def main():
    path = os.environ.get("CONV_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gauspcc_amd", "csrc", "conv_loop2_gfx950.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_conv_loop2.py -- do not edit.  gfx950 ISA of the pair-step tile loop of k_sparse_conv.\n")
        o = build()
        f.write("#define CONV_LOOP2_ASM \\\n")
        for ln in o:
            f.write('    "' + ln + '\\n" \\\n')
        f.write('    ""\n')
        print(f"CONV_LOOP2_ASM: {len(o)} lines")
        f.write("#define CONV_LOOP2_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + ', "vcc", "scc", "memory"\n')
        f.write(f"#define CONV_LOOP2_ROW_BYTES {ROWB}\n")
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
