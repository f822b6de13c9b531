#!/usr/bin/env python3
"""profiles/<round>_pmc_conv.json from the FETCH_SIZE / WRITE_SIZE sums of tools/pmc_traffic.sh (summary_conv.txt):
HBM bytes per conv launch, corrected as MI355X_MICROARCH.md (section HBM) prescribes.  Usage: make_pmc_json.py SUMMARY OUT.json"""
import json
import re
import sys

fetch = write = 0.0
disp = 0
for ln in open(sys.argv[1]):
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)\s+\(dispatches (\d+)\)", ln)
    if not m:
        continue
    if m.group(1) == "FETCH_SIZE":
        fetch += float(m.group(2))
        disp += int(m.group(3))
    else:
        write += float(m.group(2))
step = (2.0 * fetch + write) * 1024.0
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of `python3 bench.py --steps 1 --warmup 0 --cpu-sample 0` "
              "(tools/pmc_traffic.sh); raw sums over all k_sparse_conv* dispatches of one encode+decode step in the *_fetch_write.txt next to this file",
    "fetch_size_kb": fetch,
    "write_size_kb": write,
    "dispatches": disp,
    "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane loads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
    "hbm_bytes_per_launch": step / max(disp, 1),
    "hbm_bytes_per_step": step,
}
# optional: the SQ / GRBM counter passes of the same command (tools/pmc_conv2.sh summary) and the bench line of the round ->
# the other two factors of roofline.frac: how busy the matrix pipes were, and how much of what they computed was real pairs
if len(sys.argv) > 4:
    cnt = {}
    kern = None
    for ln in open(sys.argv[3]):
        m = re.match(r"^(\S.*?)\s*$", ln)
        if m and not ln.startswith(" "):
            kern = m.group(1)
        m = re.match(r"\s+(\w+)\s+([0-9.e+]+)\s+\(dispatches (\d+)\)", ln)
        if m and kern and "k_sparse_conv<255" in kern and ("true, true" in kern or "true, 1>" in kern):
            cnt[m.group(1)] = float(m.group(2))
    bench = None
    for ln in open(sys.argv[4]):
        if ln.startswith("{"):
            bench = json.loads(ln)
    if cnt.get("GRBM_GUI_ACTIVE") and cnt.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        out["mfma_busy"] = round(cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (cnt["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        out["mfma_busy_source"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) over the dispatches of k_sparse_conv<255, 1, true, 1> (the pair-step build), one encode + decode step"
    if cnt.get("SQ_INSTS_MFMA") and bench:
        # a 16-pair tile is 16 v_mfma_f32_16x16x4_f32 = 16 x 2048 flop; SQ_INSTS_MFMA counts wave instructions.  The dominant kernel's
        # share of the step's pair jobs is taken from its share of the conv time (it IS >= 92 % of both)
        flops = bench["roofline"]["algorithmic_flops_per_step"]
        out["tile_fill"] = round(flops / 2048.0 / cnt["SQ_INSTS_MFMA"] * 0.97, 4)
        out["tile_fill_source"] = "pair jobs of a step (bench line) x the pair kernel's 97 % share / SQ_INSTS_MFMA of its dispatches: real pairs per MFMA row issued"
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
