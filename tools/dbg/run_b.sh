cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b; rm -f gpurun_out/b/log.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/b/pytest.log
tail -2 gpurun_out/b/pytest.log
for r in 1 2 3; do
echo "== run $r" >> gpurun_out/b/log.txt
timeout 400 python tools/inflight_check.py 2 300 >> gpurun_out/b/log.txt 2>&1
done
echo "== 3 scenes" >> gpurun_out/b/log.txt
timeout 400 python tools/inflight_check.py 3 150 >> gpurun_out/b/log.txt 2>&1
echo "== 4 scenes" >> gpurun_out/b/log.txt
timeout 400 python tools/inflight_check.py 4 100 >> gpurun_out/b/log.txt 2>&1
grep -v amdgpu.ids gpurun_out/b/log.txt | grep -E "^==|steps:|solo" | cut -c1-200
timeout 300 python tools/fuzz_containers.py 1000000 48 0 7 2>&1 | tail -1
