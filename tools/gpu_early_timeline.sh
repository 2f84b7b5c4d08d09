#!/bin/bash
# kernel timelines of the pipelined bench with the early search off / on (cfg2 and cfg4:8): profiles/r05/timeline_early_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/early_tl; mkdir -p $O
for W in cfg2 cfg4:8; do for E in 0 1; do
  T=$(echo $W | tr ':' '_')_e$E; rm -rf $O/tl_$T
  AMR_EARLY_SEARCH=$E timeout 300 rocprofv3 --kernel-trace --stats -d $O/tl_$T -o prof --output-format csv -- python $R/bench.py --workload $W --steps 12 --warmup 3 --k1-events 0 --no-cpu-baseline --no-verify --spinup-ms 100 > $O/$T.log 2>&1
  F=$(find $O/tl_$T -name '*kernel_trace.csv' | head -1)
  python $R/tools/timeline.py $F 8 2>&1 | grep -v rocclr | head -34 > $O/timeline_early_$T.txt
  S=$(find $O/tl_$T -name '*kernel_stats.csv' | head -1); head -6 $S | cut -c1-150 > $O/kernel_stats_early_$T.csv
  rm -rf $O/tl_$T
  echo "== $T"; head -22 $O/timeline_early_$T.txt; tail -1 $O/timeline_early_$T.txt
done; done
