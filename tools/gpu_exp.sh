#!/bin/bash
# A/B of experimental environment switches on one box, pipelined benches only: gpu_exp.sh <tag> "<ENV=1 ...>" ... (one quoted env set per variant; "-" = none)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1)); [ "$envs" = "-" ] && envs=""
    for w in cfg2 cfg3 cfg5; do
      st=200; [ $w != cfg2 ] && st=50
      env $envs timeout 300 python bench.py --workload $w --steps $st --no-cpu-baseline > $O/bench_${w}_v${i}_r${rep}.log 2>&1
    done
  done
done
