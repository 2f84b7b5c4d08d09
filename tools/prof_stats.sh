#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench (5 steps); output under gpurun_out/prof_<tag>
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o prof --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}.log 2>&1
