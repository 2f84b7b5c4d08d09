#!/bin/bash
# GPU-box job (run via gpurun): parity tests (one process, uncaptured so runtime aborts are visible), bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
