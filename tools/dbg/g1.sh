cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_variants.py -x -q -m gpu -k "conv3d or bitstream_identical or conv_variant or variants" > gpurun_out/g1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g1/pytest.log
for v in "" rb128; do
  if [ -n "$v" ]; then export GAUSPCC_LIB=$PWD/gauspcc_amd/variants/libgauspcc_$v.so; fi
  echo "== variant '$v'" >> gpurun_out/g1/enc.log
  timeout 300 python tools/enc_only.py 1000000 10 >> gpurun_out/g1/enc.log 2>&1
  timeout 300 python tools/enc_only.py 1000000 10 >> gpurun_out/g1/enc.log 2>&1
  timeout 300 python tools/conv_log.py > gpurun_out/g1/convlog_${v:-main}.txt 2>&1
done
unset GAUSPCC_LIB
timeout 600 python bench.py > gpurun_out/g1/bench.json 2> gpurun_out/g1/bench.err
tail -3 gpurun_out/g1/pytest.log; cat gpurun_out/g1/enc.log; tail -1 gpurun_out/g1/bench.json
