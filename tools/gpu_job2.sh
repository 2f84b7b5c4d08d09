#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AMR_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
