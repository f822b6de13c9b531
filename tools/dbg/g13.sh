cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g13
QUIET="--skip-v0 --skip-stages --cpu-sample 0 --scenes-in-flight 0 --side-anchors 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/g13/t -o tr -- python3 bench.py $QUIET --steps 3 --warmup 2 > gpurun_out/g13/bench_traced.log 2>&1
f=$(find gpurun_out/g13/t -name "*kernel_trace.csv" | head -1)
python3 tools/dbg/level_spans.py $f 14 > gpurun_out/g13/spans.txt
for l in 2 4 6 9; do python3 tools/level_seq.py $f $l > gpurun_out/g13/seq_$l.txt; done
rm -rf gpurun_out/g13/t
cat gpurun_out/g13/spans.txt
