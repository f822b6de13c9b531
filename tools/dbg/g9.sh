cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g9
GAUSPCC_CONV_SPLIT_MAX=100000 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/g9/t -o tr -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --side-anchors 0 --skip-stages --skip-v0 --scenes-in-flight 0 > gpurun_out/g9/bench.log 2>&1
f=$(find gpurun_out/g9/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X","?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X","?"))))
rows.sort()
starts=[i for i,r in enumerate(rows) if "k_bbox" in r[2]]
seg=rows[starts[-1]:]
agg=collections.OrderedDict()
for s,e,n,g,w in seg:
    if "k_conv_products" in n or "k_conv_sum" in n or "k_sparse_conv_coop" in n:
        key=(n.split("(")[0].replace("gpcc::","").replace("void ",""), g)
        a=agg.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
for (n,g),(c,us) in agg.items(): print(f"{n:22s} grid {g:>9s}  x{c:3d}  avg {us/c:7.1f} us")
PY
rm -rf gpurun_out/g9/t
