#!/bin/bash
# kernel timeline of the pipelined bench for any workload: tools/gpu_tl_any.sh <workload> [bench args] -> gpurun_out/tl/timeline_<workload>.txt
W=$1; shift; T=$(echo $W | tr : _)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O; rm -rf $O/tr_$T
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$T -o prof --output-format csv -- python $R/bench.py --workload $W --steps 12 --warmup 3 --k1-events 0 --no-cpu-baseline --spinup-ms 100 --no-measure-traffic --device-state off "$@" > $O/log_$T.txt 2>&1
F=$(find $O/tr_$T -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $F ${NSTEPS:-6} > $O/timeline_$T.txt 2>&1
rm -rf $O/tr_$T
head -${HEAD:-40} $O/timeline_$T.txt
