#!/usr/bin/env python3
"""Generate gauspcc_amd/csrc/conv_loop3_gfx950.inc: the pair-step tile loop of k_sparse_conv on v_mfma_f32_32x32x2_f32 (gfx950 ISA, one
inline-asm block) -- round 4's variant of tools/gen_conv_loop2.py (same paired tile lists, same ring of tile headers, same register
budget, same sums in the same order).

Why.  A pair step of the 16x16x4 loop issues 32 MFMAs of 8 passes for its two 16-row tiles (per k-step one for each output half and
tile); the wave is in-order, and every MFMA <-> VALU / LDS / VMEM switch costs issue cycles.  The two tiles of
a step share their kernel offset, so their 32 rows can be the 32 COLUMNS of ONE product D^T (32 channels x 32 rows) = W^T (32 x 32)
X^T (32 x 32): 16 MFMAs of 16 passes (K = 2 each) -- the same matrix-pipe time, half the MFMA issues.  That the 32x32x2 form is the
same exact k-ascending fma chain as 16x16x4 is established (tools/ubench/mfma_order.hip).

Layouts (lane l: n = l % 32 = column = tile t = n / 16, entry e = n % 16; h = l / 32 = k parity):
    B (gathered rows)  step s multiplies logical channels 2 s + h: physical position 8 i + 4 h + el with i = (s & 1) + 2 (s >= 8),
                       el = (s >> 1) & 3 -- the lane loads the four 16-byte quads at bytes 16 h + 32 i of its row (64 B; the two
                       lanes of a row cover its 128 B) and MFMA number s takes register 4 i + el: no new feature layout.
    A (weights)        MFMA row m = PHYSICAL output channel m; a third pre-swizzled fragment per offset (network.hpp:
                       conv_weight_fragments_q): [q][lane][r] = W[2 (4 q + r) + lane / 32][logical_of(lane % 32)], four 1 KiB loads.
    D                  register 4 j + i of lane (n, h) = channel 8 j + 4 h + i of column n: four physically consecutive channels per
                       j -> four 16-byte LDS read-add-writes per lane and step at bytes 16 h + 32 j of the row (the 16x16x4 loop:
                       two per tile).
The odd run's half-empty last step cannot skip its second tile here (one product covers both): it multiplies row 0 into the dummy slot.
tiles.hip decides per block whether its runs are long enough to be paired at all (pflag), as for the 16x16x4 pair loop.
"""
import os

DX, DW = 3, 3
NX, NW = DX + 1, DW + 1
ROWB = int(os.environ.get("CONV_ASM_ROWB", "144"))
_x0 = 16
_w0 = _x0 + 16 * NX                    # X: NX sets x (tile a: 8, tile b: 8)
_c0 = _w0 + 16 * NW                    # W: NW sets x 16
_s0 = _c0 + 32                         # C: 2 parities x (a: 8, b: 8)
_m0 = _s0 + 32                         # S: 2 parities x (a: 8, b: 8)

V = dict(
    jn=_m0, on=_m0 + 2, rb=_m0 + 3, ao=_m0 + 5, bo=_m0 + 7,
    ra=(_m0 + 8, _m0 + 10),                              # [parity]
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


def X(s):
    return _x0 + 16 * s


def W(s):
    return _w0 + 16 * s


def C(p):
    return _c0 + 16 * p


def S(p):
    return _s0 + 16 * p


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mf(p, wset, xset, s):
    """MFMA number s of a step: logical channels 2 s + h of the 32 gathered rows against fragment register s"""
    i, el = (s & 1) + 2 * (1 if s >= 8 else 0), (s >> 1) & 3
    c = C(p)
    return [f"v_mfma_f32_32x32x2_f32 {vr(c, 16)}, v{W(wset) + s}, v{X(xset) + 4 * i + el}, {'0' if s == 0 else vr(c, 16)}"]


def loads_x(xset):
    b = X(xset)
    return [f"global_load_dwordx4 {vr(b + 4 * i, 4)}, v{V['ao']}, %[in]" + (f" offset:{32 * i}" if i else "") for i in range(4)]


def loads_w(wset):
    b = W(wset)
    return [f"global_load_dwordx4 {vr(b + 4 * i, 4)}, v{V['bo']}, %[w]" + (f" offset:{1024 * i}" if i else "") for i in range(4)]


def addr_x():
    return [f"v_lshl_add_u32 v{V['ao']}, v{V['jn']}, 7, v{V['goff']}"]


def addr_w():
    return [f"v_lshl_add_u32 v{V['bo']}, v{V['on']}, 12, v{V['loff']}"]


def sum_adds(p):
    s, c = S(p), C(p)
    return [f"v_pk_add_f32 {vr(s + 2 * i, 2)}, {vr(s + 2 * i, 2)}, {vr(c + 2 * i, 2)}" for i in range(8)]


def sum_writes(p):
    s, ra = S(p), V["ra"][p]
    return [f"ds_write_b128 v{ra}, {vr(s + 4 * j, 4)}" + (f" offset:{32 * j}" if j else "") for j in range(4)]


def sum_reads(p):
    s, ra = S(p), V["ra"][p]
    return [f"ds_read_b128 {vr(s + 4 * j, 4)}, v{ra}" + (f" offset:{32 * j}" if j else "") for j in range(4)]


def header_reads(du):
    # neighbour rows of the pair DX + 1 steps ahead and its offset DW + 1 steps ahead (their loads are issued in the next
    # step), output slots of the next step's pair.  Slots relative to the ring position of tile u.
    assert 24 + 2 * (DX + 1 + du) + 1 < RING_SLOTS
    ja, oa, ra = 2 * (DX + 1 + du), 2 * (DW + 1 + du), 2 * (1 + du)
    # (the per-lane bases hjb / hrb already select the lane's tile of the pair: + 64 / + 16 bytes for columns 16 .. 31)
    return [f"ds_read_b32 v{V['jn']}, v{V['hjp']} offset:{64 * ja}",
            f"ds_read_b32 v{V['on']}, v{V['hop']} offset:{4 * oa}",
            f"ds_read_u8 v{V['rb']}, v{V['hrp']} offset:{16 * ra}"]


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
    b = mf(p, ws, xs, 0)
    # the step's only VALU burst
    b += (ring_pointers() if du == 0 else [])
    # the previous step's LAST product register set is read by the adds below: a 16-pass MFMA's result needs 18 wait states before a
    # VALU read (gfx940 ISA, "XDL write VGPR -> VALU read"); at least ten instruction issues lie between (the step's leftover loads,
    # the wait, three scalar instructions, the MFMA above) -- ten more here
    b += ["s_nop 9"]
    b += sum_adds(q) + addr_x() + addr_w()
    b += [f"v_mad_u32_u24 v{V['ra'][p]}, v{V['rb']}, %[t1], v{V['accb']}"]
    mem = sum_writes(q) + sum_reads(p) + header_reads(du) + loads_x(nsx) + loads_w(nsw)
    rest = []
    for s in range(1, 16):
        rest += mf(p, ws, xs, s)
    stag = (staging_fetch(label) + staging_store(label, WINDOW + 8)) if du == 0 else []
    b += interleave(rest, mem, stag)
    o += b
    return o


def build():
    o = [
        "; ---- per-lane constants",
        f"v_and_b32 v{V['t0']}, 31, %[lane]",                       # n = column of the product: 16 t + e (tile of the pair, entry)
        f"v_lshrrev_b32 v{V['goff']}, 5, %[lane]",                  # h
        f"v_lshlrev_b32 v{V['goff']}, 4, v{V['goff']}",             # 16 h: byte offset of this lane's first channel quad in a row
        f"v_add_u32 v{V['accb']}, %[acc], v{V['goff']}",
        f"v_lshl_add_u32 v{V['hjb']}, v{V['t0']}, 2, %[hdr]",       # neighbour rows: slot (64 B) of the lane's tile = + 64 t, entry + 4 e: 4 n
        f"v_add_u32 v{V['hrb']}, %[hdr], v{V['t0']}",               # output slots: 16 B per tile, entry e: n
        f"v_add_u32 v{V['hrb']}, {RING_R}, v{V['hrb']}",
        f"v_and_b32 v{V['t0']}, 15, %[lane]",                       # (the header staging below works on 16-lane groups as in the 16x16x4 loop)
        f"v_mov_b32 v{V['hob']}, %[hdr]",
        f"v_add_u32 v{V['hob']}, {RING_O}, v{V['hob']}",
        f"v_lshlrev_b32 v{V['loff']}, 4, %[lane]",
        f"v_lshlrev_b32 v{V['sgr']}, 2, %[lane]",
        f"v_lshlrev_b32 v{V['sgo']}, 2, v{V['t0']}",
        f"v_add_u32 v{V['swj']}, %[hdr], v{V['loff']}",
        f"v_add_u32 v{V['swr']}, %[hdr], v{V['sgr']}",
        f"v_add_u32 v{V['swr']}, {RING_R}, v{V['swr']}",
        f"v_add_u32 v{V['swo']}, v{V['hob']}, v{V['sgo']}",
        f"v_mov_b32 v{V['ra'][1]}, v{V['accb']}",                   # "previous pair" of step 0: the dummy slot
        "; ---- pipeline prologue: X of pairs 0..DX-1 and W of pairs 0..DW-1 in flight; headers j(pair DX), o(pair DW), slots(pair 0) in registers",
        "s_waitcnt lgkmcnt(0)",
    ]
    for s in range(DX):
        o += [f"ds_read_b32 v{V['jn']}, v{V['hjb']} offset:{64 * 2 * s}"]
        o += [f"ds_read_b32 v{V['on']}, v{V['hob']} offset:{4 * 2 * s}"] if s < DW else []
        o += ["s_waitcnt lgkmcnt(0)"]
        o += addr_x() + loads_x(s) + (addr_w() + loads_w(s) if s < DW else [])
    o += [
        f"ds_read_b32 v{V['jn']}, v{V['hjb']} offset:{64 * 2 * DX}",
        f"ds_read_b32 v{V['on']}, v{V['hob']} offset:{4 * 2 * DW}",
        f"ds_read_u8 v{V['rb']}, v{V['hrb']}",
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
        "s_nop 15",                    # MFMA result -> VALU read needs 18 wait states after a 16-pass MFMA; nothing else separates them here
        "s_nop 7",
    ]
    o += sum_adds(1) + sum_writes(1)
    o += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return o


def main():
    path = os.environ.get("CONV_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gauspcc_amd", "csrc", "conv_loop3_gfx950.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_conv_loop3.py -- do not edit.  gfx950 ISA of the 32x32x2 pair-step tile loop of k_sparse_conv.\n")
        o = build()
        f.write("#define CONV_LOOP3_ASM \\\n")
        for ln in o:
            f.write('    "' + ln + '\\n" \\\n')
        f.write('    ""\n')
        print(f"CONV_LOOP3_ASM: {len(o)} lines")
        f.write("#define CONV_LOOP3_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + ', "vcc", "scc", "memory"\n')
        f.write(f"#define CONV_LOOP3_ROW_BYTES {ROWB}\n")
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
