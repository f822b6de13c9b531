cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ht
timeout 300 python tools/host_trace.py 1000000 0 > gpurun_out/ht/log.txt 2>&1
grep -v amdgpu.ids gpurun_out/ht/log.txt | tail -70 | cut -c1-160
