cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g10
GAUSPCC_CONV_SPLIT_MAX=100000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3d or bitstream_identical or tiny or cross" > gpurun_out/g10/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g10/pytest.log
GAUSPCC_CONV_SPLIT_MAX=100000 timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level  ?[0-9] |^\{|^enc_ms" > gpurun_out/g10/split_all.txt
timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level  ?[0-9] |^\{|^enc_ms" > gpurun_out/g10/default.txt
tail -2 gpurun_out/g10/pytest.log; cat gpurun_out/g10/split_all.txt; echo ---; cat gpurun_out/g10/default.txt
