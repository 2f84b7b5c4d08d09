#!/bin/bash
# usage: tools/gpu_diag.sh <diag ids...>   (bench each build/diag<N>.so; N=0 = the product library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in "$@"; do
  if [ "$d" = 0 ]; then unset AMR_LIB_OVERRIDE; else export AMR_LIB_OVERRIDE=$PWD/build/diag$d.so; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/diag$d.log 2>&1
done
