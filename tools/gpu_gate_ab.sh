#!/bin/bash
# The gate kernel behind the stop event of what precedes "its" K1 launch (AMR_GATE_EVENT=1, the product) against resident as
# soon as the tail stream reaches it (=0, round 4), alternating runs on one box; cfg3 also with the two-lanes-per-row search
# (build/libamrdemod_lpr2.so: tools/build_variant.sh lpr2 "-DAMR_K2R_LPR2=1").
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/gate_ab; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1].replace('.log',''), d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
run() { # name workload steps gate lib
  local L=$GRAFT_REPO_ROOT/rtlamr_amd/csrc/libamrdemod.so; [ -n "$5" ] && L=$GRAFT_REPO_ROOT/build/libamrdemod_$5.so
  AMR_GATE_EVENT=$4 AMR_LIB_OVERRIDE=$L timeout 300 python bench.py --workload $2 --no-cpu-baseline --steps $3 > $O/$1.log 2>&1; line $O/$1.log; }
for i in 1 2 3; do
  run cfg2_g0_$i cfg2 200 0; run cfg2_g1_$i cfg2 200 1
  run cfg3_g0_$i cfg3 40 0; run cfg3_g1_$i cfg3 40 1; run cfg3_lpr2_g1_$i cfg3 40 1 lpr2
done
for i in 1 2; do
  run cfg5_g0_$i cfg5 40 0; run cfg5_g1_$i cfg5 40 1
  run c8_g0_$i cfg4:8 100 0; run c8_g1_$i cfg4:8 100 1
  run c32_g0_$i cfg4:32 100 0; run c32_g1_$i cfg4:32 100 1
done
