import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_scan_lookback' in r['Kernel_Name'] or 'k_scan_single' in r['Kernel_Name']:
        g=int(r['Grid_Size'])//int(r['Workgroup_Size']) if 'Workgroup_Size' in r else int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
        d[(r['Kernel_Name'].split('(')[0][-18:],g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sorted(d): 
    v=d[k]; print(k, len(v), 'mean %.1f min %.1f max %.1f us'%(sum(v)/len(v),min(v),max(v)))
