#!/bin/bash
# parity tests + isolated (--depth 1) and pipelined bench lines with kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/walk3.log; : > $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 > gpurun_out/walk3_tests.log 2>&1; echo "tests exit $?" >> $O
for w in cfg2 cfg3 cfg5; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> $O
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/w3_$w -o prof -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --depth 1 --steps 20 --warmup 5 --spinup-ms 100 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
done
tail -3 gpurun_out/walk3_tests.log
python - <<'PY'
import json, glob, csv
for l in open('gpurun_out/walk3.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        bad=[k for k,v in j['config']['checks'].items() if 'MISMATCH' in str(v)]
        print(j['config']['workload'][:30], f"{j['value']/1e6:.3f}e6 step {j['ms_per_step']:.4f} steady {j['steady_ms_per_step']:.4f} k1 {r['k1_ms']:.4f} frac {r['frac']:.3f} search {r['search_ms']:.4f} whole {r['whole_path_frac']:.3f}", 'BAD' if bad else 'ok')
    else: print(l.strip())
for p in sorted(glob.glob('gpurun_out/w3_*/**/prof_kernel_stats.csv', recursive=True)):
    print('==', p)
    for r in list(csv.DictReader(open(p)))[:4]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} us")
PY
