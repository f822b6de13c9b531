cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/enc
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_containers.py -x -q -m gpu > gpurun_out/enc/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/enc/pytest.log
tail -3 gpurun_out/enc/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc/t -o tr -- python3 bench.py --skip-v0 --skip-stages --cpu-sample 0 --scenes-in-flight 0 --side-anchors 0 > gpurun_out/enc/bench.log 2>&1
grep -E "k_rc_encode|k_rc_decode" $(find gpurun_out/enc/t -name "*kernel_stats.csv" | head -1) | cut -c1-160
tail -1 gpurun_out/enc/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['enc_ms'], d['dec_ms'])"
rm -rf gpurun_out/enc/t
