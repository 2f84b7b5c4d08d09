#!/bin/bash
# kernel durations without overlap: --depth 1 keeps K1, K2, K3 of a batch one after the other on one stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-iso}; rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg5 cfg4:8; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${w/:/_} -o prof --output-format csv -- python $R/bench.py --workload $w --depth 1 --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 100 > $O/prof_${w/:/_}.log 2>&1
done
