#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_robust.py tests/test_gpu_single.py -q > $O/pytest_robust.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_robust.log
TAG=r05b bash tools/gpu_soak.sh ${1:-2000} 8 | tail -8
