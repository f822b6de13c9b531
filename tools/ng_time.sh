cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_neural_gaussians.py tests/test_gpu_rd_loop.py -x -q 2>&1 | tail -5; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts -o tr -- python3 tools/bench_side_paths.py 1000000 2>/dev/null | tail -1 > gpurun_out/ng_side.json; f=$(find gpurun_out/ts -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys,json
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'k_ng' in n or 'k_render' in n or 'k_preprocess' in n or 'k_assemble' in n or 'k_anchor' in n: print(n[:50], r['Calls'], r['AverageNs'], r['Percentage'])
d=json.load(open('gpurun_out/ng_side.json')); print({k:d[k] for k in d if 'rd' in k or 'neural' in k or 'generate' in k})
"; rm -rf gpurun_out/ts
