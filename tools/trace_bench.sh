#!/bin/bash
# rocprofv3 kernel trace + stats of one bench step; per-grid conv summary.  Usage: tools/trace_bench.sh OUTDIR
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o tr -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --side-anchors 0 --skip-stages --skip-v0 --skip-sizes --scenes-in-flight 0 > "$OUT/bench.log" 2>&1
f=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
python3 tools/conv_by_level.py "$f" > "$OUT/conv_by_level.txt"
cp $(find "$OUT/t" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
python3 tools/timeline.py "$f" > "$OUT/timeline.txt" 2>/dev/null; TL_WINDOW="${TL_WINDOW:-}" true; python3 tools/level_seq.py "$f" 4 > "$OUT/level_seq.txt" 2>/dev/null; python3 tools/level_seq.py "$f" 12 > "$OUT/level_seq12.txt" 2>/dev/null; python3 tools/dec_head.py "$f" 70 > "$OUT/dec_head.txt" 2>/dev/null; python3 tools/enc_head.py "$f" > "$OUT/enc_head.txt" 2>/dev/null
rm -rf "$OUT/t"
cat "$OUT/conv_by_level.txt"
head -30 "$OUT/kernel_stats.csv" | cut -c 1-160
tail -1 "$OUT/bench.log" | cut -c 1-400
