cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
timeout 900 python -m pytest tests/test_gpu_hac_codec.py tests/test_gpu_attributes.py -x -q -m gpu > gpurun_out/g5/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g5/pytest.log
timeout 900 python bench.py --cpu-sample 0 --skip-stages --scenes-in-flight 0 --steps 3 > gpurun_out/g5/bench.json 2> gpurun_out/g5/bench.err
timeout 900 python tools/bench_side_paths.py 1000000 > gpurun_out/g5/side_1m.json 2> gpurun_out/g5/side.err
tail -3 gpurun_out/g5/pytest.log; python -c "
import json
d=json.loads(open('gpurun_out/g5/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('side_paths'), indent=1)); print(d['value'], d['enc_ms'], d['dec_ms'])
s=json.loads(open('gpurun_out/g5/side_1m.json').read().strip().splitlines()[-1]); print(json.dumps({k: s[k] for k in ('attribute_loop','mlp_grid','rd_loop','torchac_shim','gaussian_coder')}, indent=1)[:2500])"
tail -3 gpurun_out/g5/bench.err gpurun_out/g5/side.err
