#!/bin/bash
# round 4: second input distribution (uniform bytes) + where every workload stands with the row K2 / gated tail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "uniform or cfg2-shard0" tests/test_gpu_parity.py -k "uniform or generator or cfg2-shard0" -m gpu -x -q > $O/pytest_g.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_g.log | tail -2
for d in synthetic uniform; do timeout 300 python bench.py --no-cpu-baseline --data $d > $O/bench_cfg2_$d.json 2> $O/bench_cfg2_$d.err; echo "cfg2 $d rc=$?"; done
cd /tmp && export TMPDIR=/tmp
for d in synthetic uniform; do
  rm -rf $GRAFT_REPO_ROOT/$O/pmc_lds_$d
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $GRAFT_REPO_ROOT/$O/pmc_lds_$d -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --k1-events 0 --no-cpu-baseline --spinup-ms 100 --depth 1 --data $d > $GRAFT_REPO_ROOT/$O/pmc_lds_$d.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json, re
out={}
for d in ('synthetic','uniform'):
    f=glob.glob(f'gpurun_out/r04/pmc_lds_{d}/**/*counter_collection.csv', recursive=True)
    if not f: print('no pmc for', d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('amr::','')[:24]
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    out[d]={k:{c:sum(v[len(v)//2:])/max(1,len(v[len(v)//2:])) for c,v in cs.items()} for k,cs in agg.items() if k.startswith('k1t')}
json.dump(out, open('gpurun_out/r04/pmc_lds_summary.json','w'), indent=1)
print(json.dumps(out, indent=1))
PY
for w in cfg3 cfg5 cfg4:8 cfg4:32 cfg4:40 cfg4:48 cfg4:56 cfg4:64 cfg4:80 cfg4:88 cfg4:96; do timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 > $O/bench_${w/:/_}.json 2> $O/bench_${w/:/_}.err; echo "$w rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_cfg*.json')):
    try:
        j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
        print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'k2',r['search_ms'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'], 'OK' if all('MISMATCH' not in v for v in j['config']['checks'].values()) else 'MISMATCH')
    except Exception as e: print(f, 'unreadable', e)
PY
