#!/bin/bash
# usage: tools/ab.sh <diag ids...>: bench each build/diag<N>.so (0 = product) twice, interleaved; prints value, ms/step, K1 ms
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for d in "$@"; do
    if [ "$d" = 0 ]; then unset AMR_LIB_OVERRIDE; else export AMR_LIB_OVERRIDE=$PWD/build/diag$d.so; fi
    python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric' | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('diag', '$d', j['value'], j['ms_per_step'], j['roofline']['k1_ms'])"
  done
done
