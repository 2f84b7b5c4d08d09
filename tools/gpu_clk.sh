#!/bin/bash
# average shader clock per kernel in the running bench: GRBM_GUI_ACTIVE (GPU clock cycles while busy) / dispatch duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; rm -rf $O/clk
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/clk -o pmc --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --k1-events 0 --no-cpu-baseline --spinup-ms 150 "$@" > $O/clk.log 2>&1
ls $O/clk/*/ 2>/dev/null | head; 
python - <<'PY'
import csv, glob, os, re, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04/clk'
ct=glob.glob(O+'/**/*counter_collection.csv', recursive=True)
kt=glob.glob(O+'/**/*kernel_trace.csv', recursive=True)
print(ct, kt)
rows=list(csv.DictReader(open(ct[0])))
print(rows[0].keys())
dur={}
if kt:
    for r in csv.DictReader(open(kt[0])): dur[r['Dispatch_Id']]=(int(r['Start_Timestamp']),int(r['End_Timestamp']))
agg=collections.defaultdict(list)
for r in rows:
    if r['Counter_Name']!='GRBM_GUI_ACTIVE': continue
    d=dur.get(r['Dispatch_Id'])
    if not d: continue
    name=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('amr::','')[:30]
    ns=d[1]-d[0]
    agg[name].append((float(r['Counter_Value']), ns))
for k,v in agg.items():
    v=v[len(v)//2:]   # steady half
    c=sum(x[0] for x in v)/len(v); ns=sum(x[1] for x in v)/len(v)
    print(f"{k:32s} n={len(v):4d} avg {ns/1e3:8.1f} us  GUI_ACTIVE {c:12.0f}  -> {c/ns:6.3f} GHz (if one counter per chip; /8 if summed over XCDs: {c/ns/8:6.3f})")
PY
