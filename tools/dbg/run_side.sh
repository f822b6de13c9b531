cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/side
timeout 1200 python tools/inflight_side.py 25 > gpurun_out/side/log.txt 2>&1; echo "rc $?" >> gpurun_out/side/log.txt
grep -v amdgpu.ids gpurun_out/side/log.txt | tail -14 | cut -c1-300
