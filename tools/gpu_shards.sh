#!/bin/bash
# every non-zero shard of the 8-GPU workloads on ONE GPU (AMR_BENCH_SHARD: the priming path of ranks 1..7),
# each line's checks.first_batch / last_timed_step against that shard's oracle golden
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/shards; mkdir -p $O
for W in cfg2 cfg3 cfg5 cfg4:8 cfg4:72; do
  for S in 1 2 3 4 5 6 7; do
    F=$O/$(echo $W | tr ':' '_')_shard$S.json
    AMR_BENCH_SHARD=$S timeout 300 python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --spinup-ms 40 > $F 2> $F.err
    echo "$W shard $S rc=$? $(python -c "
import json,sys
try:
    d=json.load(open('$F')); c=d['config']['checks']; print(d['value'], '|', c.get('first_batch','')[:60], '|', c.get('last_timed_step','')[:50])
except Exception as e: print('NO LINE', e)
")"
  done
done
