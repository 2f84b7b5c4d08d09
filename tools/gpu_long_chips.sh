cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/long; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1].replace('.log',''), d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'frac', r['frac'], 'k2', r['search_ms'])"; }
for C in 80 88 96 72; do
  timeout 300 python bench.py --workload cfg4:$C --no-cpu-baseline --no-verify --steps 100 > $O/c${C}.log 2>&1; line $O/c${C}.log
  timeout 300 python bench.py --workload cfg4:$C --no-cpu-baseline --no-verify --steps 60 --depth 1 > $O/c${C}_d1.log 2>&1; line $O/c${C}_d1.log
done
