cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/all
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/all/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/all/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/all/smoke.log 2>&1; echo "smoke rc $?"
timeout 900 python bench.py > gpurun_out/all/bench.json 2> gpurun_out/all/bench.err; echo "bench rc $?"
tail -3 gpurun_out/all/pytest.log; python3 -c "
import json
d=json.loads(open('gpurun_out/all/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['chunk_overhead_frac_at_4bpp'], d['scenes_in_flight'])
for s in d['roofline']['stages']: print(s['stage'], s['ms_per_step'])
"
