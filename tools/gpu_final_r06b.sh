#!/bin/bash
# round 6, second half: the bench lines that go with profiles/r06b (same library), the cfg4 sweep, the driver's round-end commands
cd $GRAFT_REPO_ROOT; O=gpurun_out/prof_r06b; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_200steps.log 2>&1
python bench.py --validate --no-cpu-baseline > $O/bench_validate.log 2>&1
AMR_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_1rank.log 2>&1
mkdir -p $O/cfg4; : > $O/cfg4_sweep.txt
for c in 8 32 40 48 56 64 72 80 88 96; do
  python bench.py --workload cfg4:$c --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic > $O/cfg4/bench_cfg4_$c.json 2> $O/cfg4/err_$c.txt
  python -c "
import json,sys
d=json.loads(open('$O/cfg4/bench_cfg4_$c.json').read().strip().splitlines()[-1]); r=d['roofline']; s=(d.get('device') or {}).get('before_timed_region') or {}
print($c, d['value'], d['ms_per_step'], r['k1_ms'], r['frac'], r['whole_path_frac'], s.get('sclk_mhz'))" >> $O/cfg4_sweep.txt
done
ROUND=r06b bash tools/gpu_fullsuite.sh > $O/fullsuite.txt 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-200; done; cat $O/cfg4_sweep.txt; tail -12 $O/fullsuite.txt | cut -c1-300
