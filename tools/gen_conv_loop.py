#!/usr/bin/env python3
"""Generate gauspcc_amd/csrc/conv_loop_gfx950.inc: the hand-scheduled tile loops of k_sparse_conv (gfx950 ISA,
each used as one inline-asm block).  The schedule is written down here once, with symbolic register names, so that
the unrolled copies of a step (register rings) cannot drift apart.

One wave owns R rows x all 32 output channels: 16 MFMAs per tile (two accumulator quads).  (A column-split variant
-- two waves sharing 255 rows, 8 MFMAs per tile each, 20 % better tile fill -- was built and measured 20 % SLOWER:
the loop is bound by L1 / issue work per tile, which the split doubles; it is not kept.)

Per step (one tile = 16 (output row, neighbour row) pairs of one kernel offset) the v_mfma_f32_16x16x4_f32 are issued
back to back from a ZERO accumulator; everything else is slotted into the shadow of the matrix pipe (an MFMA occupies
it for 32 cycles, a wave can issue ~5 other instructions meanwhile):
    gathered rows (A) and weight fragment (B) of the tile D = 3 steps ahead -- a tile waits for the slowest of its 16
        gathered rows, and with ~1/n_bar of all gathers being first touches of a row, nearly every tile contains an HBM
        miss.  vmcnt retires in order, so a step waits with vmcnt(6 (D-1)): the loads of the last D-1 steps stay in
        flight, this tile's (issued D steps ago) have landed
    header words from the wave's LDS ring: neighbour rows + offset of tile u+D+1, output rows of tile u+2
    LDS addresses of the PREVIOUS tile's 4 output rows, their running sums read, previous products added, written back
    loop bookkeeping / header staging (three wide loads every 16 tiles, moved into the ring 8 tiles later)
LDS operations of a wave execute in program order, so consecutive tiles may share output rows.
Registers v16.. are fixed here (declared as clobbers), the rest is the compiler's; 2 waves/SIMD allow 256 per wave.
Register tuples start on even registers (gfx90a+ requirement).
"""
import os

EXP = int(os.environ.get("CONV_ASM_EXP", "0"))   # developer experiments: 1 no sums in LDS, 2 no weight loads, 4 no gathers
D = 3                                            # issue distance of the tile loads (steps); A and B rings = D + 1 sets

V = dict(
    A=(16, 24, 32, 40),                # 4 sets x 8 regs
    B=(48, 64, 80, 96),                # 4 sets x 16 regs
    C=(112, 120),                      # c0 = C..C+3, c1 = C+4..C+7
    jn=128, on=129, r4nn=130, r4nxt=131, r4cur=132, r4prev=133,
    ao=134, bo=135, t0=136, wof=137,
    ra=138,                            # 138..141
    s=142,                             # 142..149
    stj=150, str=154, sto=155,
    accb=156, hj=157, hr=158, ho=159, goff=160, loff=161, dummy=162, sgr=163, sgo=164, swj=165, swr=166, swo=167,
)
CLOBBER_V = list(range(16, 168))
NSTEP = 4                              # lcm(D + 1, 2)
WINDOW = 6 * (D - 1)                   # tile loads that may stay in flight across a step start
LGKM_YOUNGER = int(os.environ.get("CONV_ASM_LGKM", "4"))   # LDS operations that may stay in flight across a step start


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mfma(c, a, b, first):
    return f"v_mfma_f32_16x16x4_f32 {vr(c, 4)}, v{a}, v{b}, {'0' if first else vr(c, 4)}"


def loads_a(aset):
    A = V["A"][aset]
    o = [f"global_load_dwordx4 {vr(A, 4)}, v{V['ao']}, %[in]",
         f"global_load_dwordx4 {vr(A + 4, 4)}, v{V['ao']}, %[in] offset:64"]
    return ["s_nop 0"] * 2 if EXP & 4 else o


def loads_b(bset):
    B = V["B"][bset]
    nb = 4
    o = [f"global_load_dwordx4 {vr(B + 4 * i, 4)}, v{V['bo']}, %[w]" + (f" offset:{1024 * i}" if i else "") for i in range(nb)]
    return ["s_nop 0"] * nb if EXP & 2 else o


def addr_a():
    return [f"v_lshl_add_u32 v{V['ao']}, v{V['jn']}, 7, v{V['goff']}"]


def addr_b():
    return [f"v_lshl_add_u32 v{V['bo']}, v{V['on']}, 12, v{V['wof']}"]


def row_addr(r4, ks):
    o = []
    for k in ks:
        o.append(f"v_bfe_u32 v{V['t0']}, v{r4}, {8 * k}, 8")
        o.append(f"v_lshl_add_u32 v{V['ra'] + k}, v{V['t0']}, 7, v{V['accb']}")
    return o


def sum_reads(ks):
    if EXP & 1:
        return []
    return [f"ds_read2_b32 {vr(V['s'] + 2 * k, 2)}, v{V['ra'] + k} offset1:16" for k in ks]


def sum_adds_writes(cp):
    if EXP & 1:
        return []
    o = []
    for k in range(4):
        o.append(f"v_add_f32 v{V['s'] + 2 * k}, v{V['s'] + 2 * k}, v{cp + k}")
        o.append(f"v_add_f32 v{V['s'] + 2 * k + 1}, v{V['s'] + 2 * k + 1}, v{cp + 4 + k}")
        o.append(f"ds_write2_b32 v{V['ra'] + k}, v{V['s'] + 2 * k}, v{V['s'] + 2 * k + 1} offset1:16")
    return o


def r4_rotate():
    return [f"v_mov_b32 v{V['r4prev']}, v{V['r4cur']}", f"v_mov_b32 v{V['r4cur']}, v{V['r4nxt']}"]


def r4_validate(du):
    # r4nxt <- tile u+du+1 exists ? its rows (read from the ring in the previous step) : the dummy row
    return [f"s_add_u32 %[t0], %[u], {du + 1}", "s_cmp_lt_u32 %[t0], %[nt]", "s_cselect_b64 vcc, -1, 0",
            f"v_cndmask_b32 v{V['r4nxt']}, v{V['dummy']}, v{V['r4nn']}, vcc"]


def header_reads(du):
    # neighbour rows + offset of tile u+du+D+1 (its loads are issued in the next step), output rows of tile u+du+2
    return [f"s_add_u32 %[t0], %[u], {du + D + 1}", "s_and_b32 %[t0], %[t0], 31",
            f"v_lshl_add_u32 v{V['t0']}, %[t0], 6, v{V['hj']}", f"ds_read_b32 v{V['jn']}, v{V['t0']}",
            f"v_lshl_add_u32 v{V['t0']}, %[t0], 2, v{V['ho']}", f"ds_read_b32 v{V['on']}, v{V['t0']}",
            f"s_add_u32 %[t0], %[u], {du + 2}", "s_and_b32 %[t0], %[t0], 31",
            f"v_lshl_add_u32 v{V['t0']}, %[t0], 4, v{V['hr']}", f"ds_read_b32 v{V['r4nn']}, v{V['t0']}"]


def staging_fetch(label):
    # header batch (u / 16) + 1 is fetched at the first step of batch u / 16 (u > 0) ...
    return [
        "s_and_b32 %[t0], %[u], 15",
        "s_cmp_eq_u32 %[t0], 0",
        f"s_cbranch_scc0 {label}_nofetch%=",
        "s_cmp_eq_u32 %[u], 0",
        f"s_cbranch_scc1 {label}_nofetch%=",
        "s_lshr_b32 %[t1], %[u], 4",
        "s_add_u32 %[t1], %[t1], 1",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 10, v{V['loff']}",
        f"global_load_dwordx4 {vr(V['stj'], 4)}, v{V['t0']}, %[tj]",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 8, v{V['sgr']}",
        f"global_load_dword v{V['str']}, v{V['t0']}, %[tr]",
        f"v_lshl_add_u32 v{V['t0']}, %[t1], 6, v{V['sgo']}",
        f"global_load_dword v{V['sto']}, v{V['t0']}, %[toc]",
        f"{label}_nofetch%=:",
    ]


def staging_store(label, younger_loads):
    # ... and moved into the ring eight steps later -- the half of the ring it replaces was last read before step u
    return [
        "s_and_b32 %[t0], %[u], 15",
        "s_cmp_eq_u32 %[t0], 8",
        f"s_cbranch_scc0 {label}_nostore%=",
        "s_cmp_eq_u32 %[u], 8",
        f"s_cbranch_scc1 {label}_nostore%=",
        f"s_waitcnt vmcnt({younger_loads})",          # only this step's tile loads are younger
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
        f"{label}_nostore%=:",
    ]


def step_wait(du, label):
    """The loads of the last D - 1 steps may stay in flight; everything older (this tile's) must have landed.  For D
    steps after a header fetch (issued behind the tile loads of step 0) its 3 loads are inside that window as well."""
    # LDS completes in order: the header reads this step needs are older than the 4 sum writes that close the previous step
    # (and than its staging writes, if any), so those may still be in flight
    lg = 0 if EXP & 1 else LGKM_YOUNGER
    if not 1 <= du <= D:
        return [f"s_waitcnt vmcnt({WINDOW}) lgkmcnt({lg})"]
    return ["s_and_b32 %[t0], %[u], 15",
            "s_cmp_eq_u32 %[t0], 0",
            f"s_cbranch_scc0 {label}_wa%=",
            "s_cmp_eq_u32 %[u], 0",
            f"s_cbranch_scc1 {label}_wa%=",
            f"s_waitcnt vmcnt({WINDOW + 3}) lgkmcnt({lg})",
            f"s_branch {label}_wb%=",
            f"{label}_wa%=:",
            f"s_waitcnt vmcnt({WINDOW}) lgkmcnt({lg})",
            f"{label}_wb%=:"]


def step(du):
    rset, cset = du % (D + 1), du % 2
    A, B, CC, CP = V["A"][rset], V["B"][rset], V["C"][cset], V["C"][1 - cset]
    label = f"s{du}"

    def mf(kk, first=False):
        return [mfma(CC, A + kk, B + kk, first), mfma(CC + 4, A + kk, B + 8 + kk, first)]

    nset = (du + D) % (D + 1)          # the set consumed by the previous step receives tile u+du+D
    la, lb = loads_a(nset), loads_b(nset)
    o = [f"; ---- step: tile u+{du}: A/B set {rset}, C set {cset}"] + step_wait(du, label) + addr_a() + addr_b()
    # all six tile loads in one burst behind the first MFMA pair (measured: spread one per pair 62.7, before the first
    # pair 65.0, two per pair over three pairs 66.2, this 67.1 TFLOP/s on the encoder's sets)
    o += mf(0, True) + la + lb + (staging_fetch(label) if du == 0 else [])
    o += mf(1)
    o += mf(2)
    o += mf(3) + r4_rotate() + row_addr(V["r4prev"], range(4))
    o += mf(4) + sum_reads(range(4))
    o += mf(5) + r4_validate(du) + header_reads(du)
    o += mf(6) + (staging_store(label, WINDOW + 6) if du == 0 else [])
    # LDS returns in order: the 4 sum reads are older than the 3 header reads (and than any staging write)
    o += [mfma(CC, A + 7, B + 7, False)] + ([] if EXP & 1 else ["s_waitcnt lgkmcnt(3)"]) + sum_adds_writes(CP)
    o += [mfma(CC + 4, A + 7, B + 15, False)]
    return o


def build():
    L = V
    o = [
        "; ---- per-lane constants",
        f"v_and_b32 v{L['t0']}, 15, %[lane]",                       # e
        f"v_lshrrev_b32 v{L['ra']}, 4, %[lane]",                    # g
        f"v_and_b32 v{L['ra'] + 1}, 3, v{L['t0']}",
        f"v_lshrrev_b32 v{L['ra'] + 2}, 2, v{L['t0']}",
        f"v_lshl_add_u32 v{L['ra'] + 1}, v{L['ra'] + 1}, 2, v{L['ra'] + 2}",   # col0 = 4 (e & 3) + (e >> 2)
        f"v_lshl_add_u32 v{L['accb']}, v{L['ra'] + 1}, 2, %[acc]",
        f"v_lshl_add_u32 v{L['hj']}, v{L['t0']}, 2, %[hdr]",
        f"v_lshl_add_u32 v{L['hr']}, v{L['ra']}, 2, %[hdr]",
        f"v_add_u32 v{L['hr']}, 2048, v{L['hr']}",
        f"v_mov_b32 v{L['ho']}, %[hdr]",
        f"v_add_u32 v{L['ho']}, 2560, v{L['ho']}",
        f"v_lshlrev_b32 v{L['goff']}, 4, v{L['ra']}",
        f"v_lshlrev_b32 v{L['loff']}, 4, %[lane]",
        f"v_mov_b32 v{L['wof']}, v{L['loff']}",
        f"v_mov_b32 v{L['dummy']}, %[dummy]",
        f"v_lshlrev_b32 v{L['sgr']}, 2, %[lane]",
        f"v_lshlrev_b32 v{L['sgo']}, 2, v{L['t0']}",
        f"v_add_u32 v{L['swj']}, %[hdr], v{L['loff']}",
        f"v_add_u32 v{L['swr']}, %[hdr], v{L['sgr']}",
        f"v_add_u32 v{L['swr']}, 2048, v{L['swr']}",
        f"v_add_u32 v{L['swo']}, v{L['ho']}, v{L['sgo']}",
        "; ---- pipeline prologue: A and B of tiles 0..D-1 in flight; headers j(D), o(D), r4(0), r4(1) in registers",
        "s_waitcnt lgkmcnt(0)",
        f"ds_read_b32 v{L['r4nxt']}, v{L['hr']}",
        f"v_mov_b32 v{L['r4cur']}, v{L['dummy']}",
    ]
    for t in range(D):
        o += [f"ds_read_b32 v{L['jn']}, v{L['hj']} offset:{64 * t}", f"ds_read_b32 v{L['on']}, v{L['ho']} offset:{4 * t}", "s_waitcnt lgkmcnt(0)"]
        o += addr_a() + addr_b() + loads_a(t) + loads_b(t)
    o += [
        f"ds_read_b32 v{L['jn']}, v{L['hj']} offset:{64 * D}",
        f"ds_read_b32 v{L['on']}, v{L['ho']} offset:{4 * D}",
        f"ds_read_b32 v{L['r4nn']}, v{L['hr']} offset:16",
        "s_mov_b32 %[u], 0",
        "s_waitcnt lgkmcnt(0)",        # the steps only wait for LDS operations older than a previous step's sum writes
        "conv_loop%=:",
    ]
    for du in range(NSTEP):
        o += step(du)
        if du == 1:   # leave after an even number of steps when the list is exhausted (set 1 holds the last products either way)
            o += ["s_add_u32 %[t0], %[u], 2", "s_cmp_ge_u32 %[t0], %[nt]", "s_cbranch_scc1 conv_drain%="]
    o += [
        f"s_add_u32 %[u], %[u], {NSTEP}",
        "s_cmp_lt_u32 %[u], %[nt]",
        "s_cbranch_scc1 conv_loop%=",
        "conv_drain%=:",
        "; ---- drain: products of the last step (set 1's accumulators, rows r4cur)",
    ]
    o += row_addr(L["r4cur"], range(4)) + sum_reads(range(4)) + ["s_waitcnt lgkmcnt(0)"] + sum_adds_writes(V["C"][1])
    o += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return o


def main():
    path = os.environ.get("CONV_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gauspcc_amd", "csrc", "conv_loop_gfx950.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_conv_loop.py -- do not edit.  gfx950 ISA of the k_sparse_conv tile loops.\n")
        for name in ("CONV_LOOP_ASM",):
            o = build()
            f.write(f"#define {name} \\\n")
            for ln in o:
                f.write('    "' + ln + '\\n" \\\n')
            f.write('    ""\n')
            print(f"{name}: {len(o)} lines")
        f.write("#define CONV_LOOP_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + ', "vcc", "scc", "memory"\n')
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
