cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g3/t -o tr -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --side-anchors 0 --skip-stages --skip-v0 --scenes-in-flight 0 > gpurun_out/g3/bench.log 2>&1
f=$(find gpurun_out/g3/t -name "*kernel_trace.csv" | head -1)
python3 tools/rc_by_launch.py $f > gpurun_out/g3/rc.txt
cp $(find gpurun_out/g3/t -name "*kernel_stats.csv" | head -1) gpurun_out/g3/kernel_stats.csv
python3 tools/timeline.py $f > gpurun_out/g3/timeline.txt
rm -rf gpurun_out/g3/t
cat gpurun_out/g3/rc.txt | tail -70
