cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/reg; rm -f gpurun_out/reg/log.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/reg/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/reg/pytest.log
tail -2 gpurun_out/reg/pytest.log
for r in 1 2 3; do
echo "== run $r" >> gpurun_out/reg/log.txt
timeout 400 python tools/inflight_check.py 2 300 >> gpurun_out/reg/log.txt 2>&1
done
echo "== 3 scenes" >> gpurun_out/reg/log.txt
timeout 400 python tools/inflight_check.py 3 150 >> gpurun_out/reg/log.txt 2>&1
grep -v amdgpu.ids gpurun_out/reg/log.txt | grep -E "^==|steps:|solo" | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/reg/t -o tr -- python3 bench.py --skip-v0 --skip-stages --cpu-sample 0 --scenes-in-flight 0 --side-anchors 0 > gpurun_out/reg/bench.log 2>&1
grep -E "k_rc_encode|k_rc_decode" $(find gpurun_out/reg/t -name "*kernel_stats.csv" | head -1) | cut -c1-60,140-200
rm -rf gpurun_out/reg/t
timeout 600 python bench.py --skip-v0 --cpu-sample 0 --side-anchors 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['frac'], d['scenes_in_flight'])
for s in d['roofline']['stages']: print(s['stage'], s['ms_per_step'])
"
