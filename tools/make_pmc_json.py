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
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
