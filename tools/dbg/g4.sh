cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_containers.py tests/test_gpu_conv_variants.py -x -q -m gpu > gpurun_out/g4/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g4/pytest.log
timeout 300 python tools/conv_log.py > gpurun_out/g4/convlog.txt 2>&1
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/g4/bench.json 2> gpurun_out/g4/bench.err
tail -4 gpurun_out/g4/pytest.log; grep -E "^dec level|^enc|^\{" gpurun_out/g4/convlog.txt; tail -1 gpurun_out/g4/bench.json | cut -c1-900
