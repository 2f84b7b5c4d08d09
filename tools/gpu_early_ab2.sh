#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/early; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1], d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
for i in 1 2; do for C in 8 32 40 48; do for E in 0 1; do
  AMR_EARLY_SEARCH=$E timeout 300 python bench.py --workload cfg4:$C --steps 100 --no-cpu-baseline > $O/c${C}_e${E}_$i.log 2>&1; line $O/c${C}_e${E}_$i.log
done; done; done
