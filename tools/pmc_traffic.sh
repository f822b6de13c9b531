#!/bin/bash
# HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
# (they do not fit one pass on gfx950; never combined with tracing).  Usage: tools/pmc_traffic.sh OUTDIR
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python3 bench.py --steps 1 --warmup 0 ${BENCH_ARGS:---cpu-sample 0 --side-anchors 0} > "$OUT/$C.log" 2>&1
  echo "pass $C exit $?"
done
python3 tools/pmc_summary.py "$OUT" k_sparse_conv > "$OUT/summary_conv.txt"
cat "$OUT/summary_conv.txt"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
