#!/bin/bash
cd $GRAFT_REPO_ROOT
AMR_SERIAL_TAIL=1 bash tools/gpu_timeline.sh serial
AMR_SERIAL_TAIL=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_serial.json 2>/dev/null; tail -c 700 gpurun_out/r04/bench_serial.json
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_split.json 2>/dev/null; tail -c 700 gpurun_out/r04/bench_split.json
