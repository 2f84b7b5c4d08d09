#!/bin/bash
# A/B of an environment switch on ONE box: isolated kernel durations (--depth 1) and pipelined benches, alternating
# usage: gpu_ab.sh <tag> <ENVVAR> [workloads...]
TAG=$1; VAR=$2; shift 2; WLS=${@:-cfg2 cfg3 cfg5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for w in $WLS; do
  for v in a b; do
    if [ $v = b ]; then export $VAR=1; else unset $VAR; fi
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_iso_${w}_${v}${rep} -o prof --output-format csv -- python $R/bench.py --workload $w --depth 1 --steps 20 --warmup 5 --no-cpu-baseline --no-verify --spinup-ms 100 > $O/prof_iso_${w}_${v}${rep}.log 2>&1
    (cd $R && timeout 300 python bench.py --workload $w --steps 50 --no-cpu-baseline > $O/bench_${w}_${v}${rep}.log 2>&1)
  done
done
done
unset $VAR
