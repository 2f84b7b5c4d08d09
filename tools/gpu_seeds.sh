#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
AMR_STALE_DBG=1 timeout 600 python tools/debug_seed.py 2233 > $O/seeds.txt 2>&1
head -60 $O/seeds.txt
