cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8; rm -f gpurun_out/g8/enc.log
for p in 100 125 150 175 200 250; do
  echo "== PAIR_MIN=$p" >> gpurun_out/g8/enc.log
  GAUSPCC_CONV_PAIR_MIN=$p timeout 300 python tools/enc_only.py 1000000 10 2>/dev/null >> gpurun_out/g8/enc.log
done
GAUSPCC_CONV_PAIR_MIN=125 timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level 1[1-4]|^\{" >> gpurun_out/g8/enc.log
GAUSPCC_CONV_PAIR_MIN=200 timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level 1[1-4]|^\{" >> gpurun_out/g8/enc.log
cat gpurun_out/g8/enc.log
