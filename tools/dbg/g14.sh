cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g14
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_containers.py tests/test_gpu_conv_variants.py -x -q -m gpu > gpurun_out/g14/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g14/pytest.log
timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level|^enc|^\{" > gpurun_out/g14/convlog.txt
GAUSPCC_COOP_TALL=0 timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level|^enc|^\{" > gpurun_out/g14/convlog_off.txt
tail -3 gpurun_out/g14/pytest.log; cut -c1-150 gpurun_out/g14/convlog.txt | sed -n 1,16p;  cut -c1-150 gpurun_out/g14/convlog_off.txt | sed -n 1,2p
