#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
bash tools/gpu_ab.sh ab_k3m AMR_K3_MERGE cfg2 cfg3 cfg5
