#!/bin/bash
# the side-path part of tools/refresh_profiles.sh alone (attribute loops, Gaussian coder, RD loop; kernel stats; k_render counters)
set -u
OUT=$1
R=${2:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout 900 python3 tools/bench_side_paths.py 1000000 2> "$OUT/side.err" | tail -1 > "$OUT/${R}_side_paths.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ts" -o tr -- python3 tools/bench_side_paths.py 1000000 > /dev/null 2>&1
cp $(find "$OUT/ts" -name "*kernel_stats.csv" | head -1) "$OUT/${R}_side_kernel_stats.csv"; rm -rf "$OUT/ts"
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_r" -o pmc -- python3 tools/bench_side_paths.py 1000000 > /dev/null 2>&1
python3 tools/pmc_summary.py "$OUT/sq_r" "k_render" > "$OUT/${R}_render_counters.txt"; python3 tools/pmc_summary.py "$OUT/sq_r" "k_preprocess" >> "$OUT/${R}_render_counters.txt"
python3 tools/pmc_summary.py "$OUT/sq_r" "k_ng_emit" >> "$OUT/${R}_render_counters.txt"; python3 tools/pmc_summary.py "$OUT/sq_r" "k_ng_opacity" >> "$OUT/${R}_render_counters.txt"
rm -rf "$OUT/sq_r"
python3 -c "
import json; d=json.load(open('$OUT/${R}_side_paths.json')); print(d['attribute_loop_hac_plus'].get('repetitions')); print(d['rd_loop'])"
