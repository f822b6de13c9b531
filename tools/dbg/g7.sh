cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_variants.py -x -q -m gpu -k "conv3d or bitstream_identical or conv_variant or variants or roundtrip_1m" > gpurun_out/g7/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g7/pytest.log
for p in 1 0; do
  echo "== GAUSPCC_CONV_PAIR=$p" >> gpurun_out/g7/enc.log
  GAUSPCC_CONV_PAIR=$p timeout 300 python tools/enc_only.py 1000000 10 2>/dev/null >> gpurun_out/g7/enc.log
  GAUSPCC_CONV_PAIR=$p timeout 300 python tools/enc_only.py 1000000 10 2>/dev/null >> gpurun_out/g7/enc.log
done
timeout 300 python tools/conv_log.py > gpurun_out/g7/convlog.txt 2>&1
tail -4 gpurun_out/g7/pytest.log; cat gpurun_out/g7/enc.log; grep -E "^enc|^dec level 1[0-4]|^\{" gpurun_out/g7/convlog.txt
