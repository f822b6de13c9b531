cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_containers.py tests/test_gpu_robustness.py tests/test_gpu_conv_variants.py -x -q -m gpu > gpurun_out/q/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/q/pytest.log
tail -3 gpurun_out/q/pytest.log
for i in 1 2 3; do timeout 600 python bench.py --skip-v0 --skip-stages --cpu-sample 0 --side-anchors 0 --scenes-in-flight 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['enc_ms'], d['dec_ms'], d['roofline']['frac'])
"; done
