cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_rd_loop.py -x -q 2>&1 | tail -2; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts -o tr -- python3 tools/bench_side_paths.py 1000000 2>/dev/null | tail -1 > gpurun_out/r05_side_c.json; f=$(find gpurun_out/ts -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys,json
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'k_render' in n or 'k_preprocess' in n: print(n[:40], r['Calls'], r['AverageNs'], r['Percentage'])
print(json.load(open('gpurun_out/r05_side_c.json'))['rd_loop'])
"; rm -rf gpurun_out/ts
