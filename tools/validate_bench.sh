#!/bin/bash
# On the GPU box: the default bench line, the same with --validate, and a kernel-stats profile of the latter.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/validate; rm -rf $O; mkdir -p $O
python $R/bench.py --no-cpu-baseline > $O/plain.log 2>&1
python $R/bench.py --no-cpu-baseline --validate > $O/validate.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o prof --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --validate > $O/stats.log 2>&1
tail -1 $O/plain.log; tail -1 $O/validate.log; head -14 $O/stats/prof_kernel_stats.csv
