#!/bin/bash
# Everything profiles/ holds for a round, measured in one go on the GPU box.  Usage: tools/refresh_profiles.sh OUTDIR [ROUND]
# (every step under its own `timeout`; PMC passes never combined with tracing)
set -u
OUT=$1
R=${2:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
QUIET="--skip-v0 --skip-stages --skip-sizes --cpu-sample 0 --scenes-in-flight 0 --side-anchors 0"
# 1. HBM traffic of the conv kernels: FETCH_SIZE / WRITE_SIZE in separate passes
BENCH_ARGS="$QUIET" bash tools/pmc_traffic.sh "$OUT/pmc" > "$OUT/pmc_traffic.log" 2>&1
cp "$OUT/pmc/summary_conv.txt" "$OUT/${R}_pmc_conv_fetch_write.txt"
python3 tools/make_pmc_json.py "$OUT/${R}_pmc_conv_fetch_write.txt" "$OUT/${R}_pmc_conv.json" > /dev/null
cp "$OUT/${R}_pmc_conv.json" profiles/${R}_pmc_conv.json     # bench.py reads the per-launch traffic from here
# 1b. SQ / TCP / GRBM counters of the conv kernels; with a bench line they give roofline.mfma_busy / tile_fill
BENCH_ARGS="$QUIET" bash tools/pmc_conv2.sh "$OUT/pmc2" > "$OUT/pmc_counters.log" 2>&1
cp "$OUT/pmc2/summary.txt" "$OUT/${R}_pmc_conv_counters.txt"; rm -rf "$OUT/pmc2"
timeout 600 python3 bench.py $QUIET > "$OUT/bench_quiet.json" 2> /dev/null
python3 tools/make_pmc_json.py "$OUT/${R}_pmc_conv_fetch_write.txt" "$OUT/${R}_pmc_conv.json" "$OUT/${R}_pmc_conv_counters.txt" "$OUT/bench_quiet.json" > /dev/null
cp "$OUT/${R}_pmc_conv.json" profiles/${R}_pmc_conv.json
# 2. the bench line (default flags) and the same command under the kernel tracer
timeout 600 python3 bench.py > "$OUT/${R}_bench.json" 2> "$OUT/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o tr -- python3 bench.py --steps 5 --warmup 1 $QUIET > "$OUT/bench_traced.log" 2>&1
cp $(find "$OUT/t" -name "*kernel_stats.csv" | head -1) "$OUT/${R}_kernel_stats.csv"
f=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" > "$OUT/${R}_timeline.txt"
python3 tools/conv_by_level.py "$f" > "$OUT/${R}_conv_by_level.txt"
python3 tools/level_breakdown.py "$f" > "$OUT/${R}_decode_levels.txt" 2>&1
rm -rf "$OUT/t" "$OUT/pmc"
# 2a. K scenes through one chain of launches (gpcc_encode_batch / gpcc_decode_batch): timelines of the three batch shapes of bench.py's `batched` object
for cfg in "8 100000" "32 10000" "2 1000000"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/bt" -o tr -- python3 tools/batch_probe.py $1 $2 4 > "$OUT/batch_probe_$1x$2.log" 2>&1
  f=$(find "$OUT/bt" -name "*kernel_trace.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace -- python3 tools/batch_probe.py $1 $2 4   (last encode + decode of the batch; tools/timeline.py)"; grep "^iter" "$OUT/batch_probe_$1x$2.log" | sed 's/^/# /'; python3 tools/timeline.py "$f"; } > "$OUT/${R}_batch_timeline_$1x$2.txt"
  rm -rf "$OUT/bt"
done
# 2b. convolutions per level (per-launch HIP events inside the library)
{ echo "# tools/conv_log.py 1000000 (per-launch HIP events, GAUSPCC_CONV_LOG=1): convolutions of one encode + decode of the 1 M-point bench cloud, per level"
  echo "# 'enc level 0 / 1' = the prior set (levels 0..L-2) / the target set (levels 1..L-1), all levels in one launch; 'dec level g' = the launches on level g's nodes"
  timeout 300 python3 tools/conv_log.py 1000000 2>/dev/null | grep -E "^enc|^dec|^\{"; } > "$OUT/${R}_conv_levels.txt"
# 4. the callers either side of the path (attribute loop, Gaussian coder, generate_neural_gaussians + rasteriser)
timeout 600 python3 tools/bench_side_paths.py 1000000 2> "$OUT/side.err" | tail -1 > "$OUT/${R}_side_paths.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ts" -o tr -- python3 tools/bench_side_paths.py 1000000 > /dev/null 2>&1
cp $(find "$OUT/ts" -name "*kernel_stats.csv" | head -1) "$OUT/${R}_side_kernel_stats.csv"; rm -rf "$OUT/ts"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/sp_$C" -o pmc -- python3 tools/bench_side_paths.py 1000000 > /dev/null 2>&1
done
python3 tools/pmc_summary.py "$OUT" "k_" > "$OUT/${R}_side_pmc_fetch_write.txt"
rm -rf "$OUT"/sp_*
# 4b. the rasteriser's compute side: VALU instructions and busy cycles of k_render / k_preprocess (one pass)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_r" -o pmc -- python3 tools/bench_side_paths.py 1000000 > /dev/null 2>&1
python3 tools/pmc_summary.py "$OUT/sq_r" "k_render" > "$OUT/${R}_render_counters.txt"; python3 tools/pmc_summary.py "$OUT/sq_r" "k_preprocess" >> "$OUT/${R}_render_counters.txt"
python3 tools/pmc_summary.py "$OUT/sq_r" "k_ng_emit" >> "$OUT/${R}_render_counters.txt"; python3 tools/pmc_summary.py "$OUT/sq_r" "k_ng_opacity" >> "$OUT/${R}_render_counters.txt"
rm -rf "$OUT/sq_r"
# 4c. the conv tail-packing gate (round 6): the 4x4x1_16B micro-benchmark (the list-length model is a CPU script: tools/tile_fill_model.py)
[ -x tools/ubench/mfma_tail ] && timeout 120 ./tools/ubench/mfma_tail > "$OUT/${R}_mfma_tail.txt" 2>&1
# 4d. the persistent-class boundary (round 6): batches and the solo scene with merged levels of up to 40 k / 64 k nodes on the persistent launches
{ for fm in 16384 40000 65536; do for cfg in "8 100000" "32 10000" "2 1000000"; do echo "== GAUSPCC_FUSED_MAX=$fm  batch_probe $cfg"; GAUSPCC_DEV=1 GAUSPCC_FUSED_MAX=$fm timeout 200 python3 tools/batch_probe.py $cfg 4 2>&1 | grep -E "^iter [23]|kernels per"; done; done; } > "$OUT/${R}_batch_fused_max.txt" 2>&1
# 5. the grid-barrier / launch-chain microbenchmark behind the small-level fusion decision (built by tools/build_variants.sh or by hand)
[ -x tools/ubench/grid_sync ] && timeout 300 ./tools/ubench/grid_sync > "$OUT/${R}_grid_sync.txt" 2>&1
tail -1 "$OUT/${R}_bench.json"
