#!/bin/bash
# early search on / off: the GPU suite with it on, then alternating bench runs on the same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/early; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_early.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_early.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']['checks']
ok=all('MISMATCH' not in str(v) for v in c.values())
print('$1'.split('/')[-1], d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'fill', d['pipeline_fill_ms'], 'k1', r['k1_ms'], 'k2', r['search_ms'], 'checks', 'ok' if ok else c)"; }
for i in 1 2 3; do
  for E in 0 1; do
    AMR_EARLY_SEARCH=$E timeout 300 python bench.py --no-cpu-baseline > $O/b200_e${E}_$i.log 2>&1; line $O/b200_e${E}_$i.log
  done
done
for E in 0 1; do AMR_EARLY_SEARCH=$E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_e${E}.log 2>&1; line $O/b20_e${E}.log; done
for E in 0 1; do AMR_EARLY_SEARCH=$E timeout 300 python bench.py --workload cfg4:32 --steps 50 --no-cpu-baseline > $O/c32_e${E}.log 2>&1; line $O/c32_e${E}.log; done
