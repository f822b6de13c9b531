cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/g12/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g12/pytest.log
timeout 900 python bench.py > gpurun_out/g12/bench.json 2> gpurun_out/g12/bench.err
tail -3 gpurun_out/g12/pytest.log; python3 -c "
import json
d=json.loads(open('gpurun_out/g12/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['chunk_overhead_frac_at_4bpp'], d['chunk_overhead_bytes'], d['container_bytes'])
"
