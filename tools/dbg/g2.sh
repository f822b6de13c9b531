cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_containers.py -x -q -m gpu > gpurun_out/g2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g2/pytest.log
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/g2/bench.json 2> gpurun_out/g2/bench.err
timeout 600 python bench.py --cpu-sample 0 --chunk-log2 10 --skip-stages --scenes-in-flight 0 > gpurun_out/g2/bench_cl10.json 2>> gpurun_out/g2/bench.err
tail -15 gpurun_out/g2/pytest.log; tail -1 gpurun_out/g2/bench.json; tail -1 gpurun_out/g2/bench_cl10.json; tail -5 gpurun_out/g2/bench.err
