#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline"
for c in 72 80 88 96 8; do timeout 300 $B --workload cfg4:$c --steps 50 > $O/bench_cfg4_$c.log 2>&1; done
