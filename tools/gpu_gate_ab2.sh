cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/gate_ab; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1].replace('.log',''), d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'])"; }
run() { AMR_GATE_EVENT=$4 timeout 300 python bench.py --workload $2 --no-cpu-baseline --no-verify --steps $3 > $O/$1.log 2>&1; line $O/$1.log; }
for i in 1 2 3 4; do
  for C in 32 48 56; do run c${C}_g0_b$i cfg4:$C 150 0; run c${C}_g1_b$i cfg4:$C 150 1; done
done
