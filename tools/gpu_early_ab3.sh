#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/early; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1], d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
for i in 1 2 3; do for E in 0 1; do
  AMR_EARLY_SEARCH=$E timeout 300 python bench.py --workload cfg4:8 --steps 100 --no-cpu-baseline > $O/q_c8_e${E}_$i.log 2>&1; line $O/q_c8_e${E}_$i.log
done; done
AMR_EARLY_SEARCH=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/q_cfg2_e1.log 2>&1; line $O/q_cfg2_e1.log
