#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-300) > gpurun_out/dbg.log 2>&1
(timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 0 2>&1 | tail -3 | cut -c1-300) >> gpurun_out/dbg.log 2>&1
(timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --depth 1 2>&1 | tail -3 | cut -c1-300) >> gpurun_out/dbg.log 2>&1
(AMR_EXP_NOWAIT=1 timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-300) >> gpurun_out/dbg.log 2>&1
cat gpurun_out/dbg.log
