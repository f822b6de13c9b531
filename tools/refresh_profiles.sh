#!/bin/bash
# Everything profiles/ holds for a round, measured in one go on the GPU box.  Usage: tools/refresh_profiles.sh OUTDIR
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
bash tools/pmc_traffic.sh "$OUT/pmc" > "$OUT/pmc_traffic.log" 2>&1
cp "$OUT/pmc/summary_conv.txt" "$OUT/pmc_conv_fetch_write.txt"
python3 tools/make_pmc_json.py "$OUT/pmc_conv_fetch_write.txt" "$OUT/pmc_conv.json" > /dev/null
cp "$OUT/pmc_conv.json" profiles/r01_pmc_conv.json     # bench.py reads the per-launch traffic from here
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o tr -- python3 bench.py --cpu-sample 0 > "$OUT/bench_traced.log" 2>&1
cp $(find "$OUT/t" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
f=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" > "$OUT/timeline.txt"
python3 tools/conv_by_level.py "$f" > "$OUT/conv_by_level.txt"
rm -rf "$OUT/t" "$OUT/pmc"
BENCH_ARGS="" bash tools/pmc_conv2.sh "$OUT/pmc2" > "$OUT/pmc_counters.log" 2>&1
cp "$OUT/pmc2/summary.txt" "$OUT/pmc_conv_counters.txt"; rm -rf "$OUT/pmc2"
tail -1 "$OUT/bench.json"
