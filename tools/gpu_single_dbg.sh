#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
AMR_SINGLE_DBG=1 timeout 300 python tools/single_block_rate.py > $O/single_block_dbg.txt 2>&1; grep -v amdgpu.ids $O/single_block_dbg.txt | tail -12
