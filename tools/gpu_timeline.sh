#!/bin/bash
# kernel timeline of the pipelined bench (cfg2, depth 3) and its stats: gpurun_out/r04/timeline_<tag>.txt
TAG=${1:-a}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; rm -rf $O/tl_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tl_$TAG -o prof --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --k1-events 0 --no-cpu-baseline --spinup-ms 150 "$@" > $O/tl_$TAG.log 2>&1
F=$(find $O/tl_$TAG -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $F 5 > $O/timeline_$TAG.txt 2>&1
S=$(find $O/tl_$TAG -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats_$TAG.csv
tail -40 $O/timeline_$TAG.txt; head -12 $O/kernel_stats_$TAG.csv
rm -rf $O/tl_$TAG
