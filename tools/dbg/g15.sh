cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g15
GAUSPCC_CONV_TALL_MIN=16400 timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level|^enc|^\{" > gpurun_out/g15/convlog_tm.txt
cut -c1-150 gpurun_out/g15/convlog_tm.txt | sed -n 10,14p
