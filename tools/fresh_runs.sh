#!/bin/bash
# tools/fresh_runs.sh [n]: the driver's bench command in n fresh processes on one box (first allocation each time):
# one line per run -> gpurun_out/fresh_runs.txt (copied into profiles/<round>/ by hand with the box noted)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/fresh_runs.txt; : > $O
N=${1:-10}
for i in $(seq 1 $N); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('run %2d  value %9.0f Msamples/s  ms_per_step %.4f  steady %.4f  K1 %.4f ms  frac %.3f  checks %s' % ($i, d['value'], d['ms_per_step'], d['steady_ms_per_step'], r['k1_ms'], r['frac'], 'ok' if 'golden' in d['config']['checks']['hit_count'] else d['config']['checks']))" >> $O
done
cat $O
