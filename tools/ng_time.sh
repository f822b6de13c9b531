# kernel table of the RD loop (generate_neural_gaussians + rasteriser) at 1 M anchors: tools/bench_side_paths.py under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts -o tr -- python3 tools/bench_side_paths.py 1000000 2>/dev/null | tail -1 > gpurun_out/ng_side.json; f=$(find gpurun_out/ts -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys,json
rows=list(csv.DictReader(open('$f')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:28] + [r for r in rows[28:] if 'k_ng' in r['Name'] or 'k_render' in r['Name']]: print('%-60s %6s %12.1f %8.2f'%(r['Name'][:60], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
d=json.load(open('gpurun_out/ng_side.json')); print(d.get('rd_loop'))
"; rm -rf gpurun_out/ts
