#!/bin/bash
# the wave-per-block K1 (k1_coop.h): its tests, then the small-batch sweep and the one-block loop through the regular kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_K:-} > $O/pytest_coop.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_coop.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python tools/coop_sweep.py > $O/coop_sweep.txt 2>&1; grep -v amdgpu.ids $O/coop_sweep.txt | tail -16
AMR_NO_SINGLE=1 timeout 300 python tools/single_block_rate.py > $O/single_block_regular.txt 2>&1; grep -v amdgpu.ids $O/single_block_regular.txt | tail -4
