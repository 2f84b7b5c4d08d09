#!/bin/bash
# the GPU suite, then an A/B of build/lib_head.so against the product library (WLS = workloads)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 2>&1 | tail -3
WLS="${WLS:-cfg5 cfg3 cfg2}" tools/gpu_libab.sh ab_k3 build/lib_head.so - > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/r3_summary.py gpurun_out/ab_k3 | grep -v 'synth\|hist_update\|k1t_demod\|k_done\|k4_r900\|dense'
