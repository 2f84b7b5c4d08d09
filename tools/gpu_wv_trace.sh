#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gate_mode; mkdir -p $O; rm -rf $O/tr
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o prof --output-format csv -- python $R/bench.py --workload cfg3 --steps 10 --warmup 3 --k1-events 0 --no-cpu-baseline --spinup-ms 50 --no-measure-traffic --device-state off > $O/trace_cfg3_value.log 2>&1
S=$(find $O/tr -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats_cfg3_value.csv
F=$(find $O/tr -name '*kernel_trace.csv' | head -1); python $R/tools/timeline.py $F 3 > $O/timeline_cfg3_value.txt 2>&1
rm -rf $O/tr
cut -c1-150 $O/kernel_stats_cfg3_value.csv | head -14; head -40 $O/timeline_cfg3_value.txt
