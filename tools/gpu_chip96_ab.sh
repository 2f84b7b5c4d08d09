#!/bin/bash
# chip 96 through the tile kernel: the product's configuration (NW 8, NLC 1, HDL 40: 247 registers) against NW 8, NLC 3, HDL 32
# (256 registers: build/libamrdemod_c96b.so) and the first-generation kernel (build/libamrdemod_walk.so, built before the change)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/c96; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1].replace('.log',''), d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'frac', r['frac'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
for i in 1 2 3; do
  for V in ${VARIANTS:-prod c96b walk}; do
    L=$GRAFT_REPO_ROOT/rtlamr_amd/csrc/libamrdemod.so; [ $V != prod ] && L=$GRAFT_REPO_ROOT/build/libamrdemod_$V.so
    AMR_LIB_OVERRIDE=$L timeout 300 python bench.py --workload ${W:-cfg4:96} --no-cpu-baseline --steps 100 > $O/${V}_$i.log 2>&1; line $O/${V}_$i.log
  done
done
