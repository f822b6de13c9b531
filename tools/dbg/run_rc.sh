cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rc; rm -f gpurun_out/rc/log.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_containers.py tests/test_gpu_robustness.py -x -q -m gpu > gpurun_out/rc/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/rc/pytest.log
tail -3 gpurun_out/rc/pytest.log
for r in 1 2 3; do
echo "== run $r" >> gpurun_out/rc/log.txt
timeout 600 python tools/inflight_check.py 2 300 >> gpurun_out/rc/log.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/rc/log.txt | cut -c1-300 | tail -8
timeout 600 python bench.py --skip-v0 --cpu-sample 0 --side-anchors 0 2>gpurun_out/rc/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['frac'], d['scenes_in_flight'])
for s in d['roofline']['stages']: print(s['stage'], s['ms_per_step'])
"
