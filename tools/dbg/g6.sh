cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g6
for v in "" w3; do
  if [ -n "$v" ]; then export GAUSPCC_LIB=$PWD/gauspcc_amd/variants/libgauspcc_$v.so; fi
  echo "== variant '$v'" >> gpurun_out/g6/enc.log
  timeout 300 python tools/enc_only.py 1000000 10 2>/dev/null >> gpurun_out/g6/enc.log
  timeout 300 python tools/enc_only.py 1000000 10 2>/dev/null >> gpurun_out/g6/enc.log
done
cat gpurun_out/g6/enc.log
