#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --workload cfg5 --steps 4 --warmup 1 --spinup-ms 0 --no-cpu-baseline --write-golden > gpurun_out/gold_cfg5.log 2>&1; tail -2 gpurun_out/gold_cfg5.log | cut -c1-1500
cp tests/golden/bench_counts.json gpurun_out/bench_counts.json
