#!/bin/bash
# the one-launch single-block path: its tests, the robustness tests, the latency figure (python loop and tools/single_block_rate.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_single.py tests/test_gpu_robust.py tests/test_gpu_capture.py -q > $O/pytest_single.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_single.log
AMR_SINGLE_DBG=1 timeout 300 python tools/single_block_rate.py > $O/single_block.txt 2>&1; grep -v amdgpu.ids $O/single_block.txt | tail -8
AMR_NO_SINGLE=1 timeout 300 python tools/single_block_rate.py > $O/single_block_regular.txt 2>&1; grep -v amdgpu.ids $O/single_block_regular.txt | tail -4
