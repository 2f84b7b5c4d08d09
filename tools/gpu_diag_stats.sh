#!/bin/bash
# usage: tools/gpu_diag_stats.sh <diag ids...>  -> rocprofv3 kernel stats of bench for each build/diag<N>.so (0 = product)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in "$@"; do
  if [ "$d" = 0 ]; then unset AMR_LIB_OVERRIDE; else export AMR_LIB_OVERRIDE=$R/build/diag$d.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ds_$d -o prof --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ds_$d.log 2>&1
done
