#!/bin/bash
# A/B of library builds on one box: gpu_libab.sh <tag> <lib paths relative to the repo ...>   ("-" = the product library)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    i=$((i+1))
    if [ "$lib" = "-" ]; then unset AMR_LIB_OVERRIDE; else export AMR_LIB_OVERRIDE=$R/$lib; fi
    for w in ${WLS:-cfg2 cfg3}; do
      st=200; [ $w != cfg2 ] && st=50
      timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_iso_${w}_v${i}_r${rep} -o prof --output-format csv -- python $R/bench.py --workload $w --depth 1 --steps 20 --warmup 5 --no-cpu-baseline --no-verify --spinup-ms 100 > $O/prof_iso_${w}_v${i}_r${rep}.log 2>&1
      (cd $R && timeout 300 python bench.py --workload $w --steps $st --no-cpu-baseline > $O/bench_${w}_v${i}_r${rep}.log 2>&1)
    done
  done
done
