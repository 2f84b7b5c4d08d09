#!/bin/bash
# cfg3 with the two-lanes-per-row search (build/libamrdemod_lpr2.so: -DAMR_K2R_LPR2=1 -DAMR_DBG_K1_DELAY=1), with an idle stretch in
# front of the batch's first K1 launch, against the product's walk: what does the slow first K1 round need?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/lpr2; mkdir -p $O
ALT=$GRAFT_REPO_ROOT/build/libamrdemod_lpr2.so
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1], d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
for i in 1 2; do
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 40 > $O/walk_$i.log 2>&1; line $O/walk_$i.log
  AMR_LIB_OVERRIDE=$ALT timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 40 > $O/lpr2_$i.log 2>&1; line $O/lpr2_$i.log
  for D in 5 20 100; do
    AMR_DBG_K1_DELAY_US=$D AMR_LIB_OVERRIDE=$ALT timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 40 > $O/lpr2_d${D}_$i.log 2>&1; line $O/lpr2_d${D}_$i.log
  done
  AMR_DBG_K1_DELAY_US=20 AMR_LIB_OVERRIDE=$ALT timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 100 > $O/cfg2_d20_$i.log 2>&1; line $O/cfg2_d20_$i.log
done
