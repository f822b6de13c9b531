cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g11
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/g11/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g11/pytest.log
timeout 300 python tools/conv_log.py 2>&1 | grep -E "^dec level|^enc|^\{" > gpurun_out/g11/convlog.txt
timeout 900 python bench.py > gpurun_out/g11/bench.json 2> gpurun_out/g11/bench.err
tail -3 gpurun_out/g11/pytest.log; head -8 gpurun_out/g11/convlog.txt | cut -c1-150; python3 -c "
import json
d=json.loads(open('gpurun_out/g11/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['chunk_overhead_frac_at_4bpp'])
for s in d['roofline']['stages']: print(s['stage'], s['ms_per_step'], s['frac'])
print(d['cpu_baseline'])
print(d['scenes_in_flight'])
"
