cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seq
QUIET="--skip-v0 --skip-stages --cpu-sample 0 --scenes-in-flight 0 --side-anchors 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq/t -o tr -- python3 bench.py $QUIET --steps 3 --warmup 2 > gpurun_out/seq/bench_traced.log 2>&1
f=$(find gpurun_out/seq/t -name "*kernel_trace.csv" | head -1)
for l in 3 5 7; do python3 tools/level_seq.py $f $l > gpurun_out/seq/seq_$l.txt; done
rm -rf gpurun_out/seq/t
cat gpurun_out/seq/seq_5.txt
