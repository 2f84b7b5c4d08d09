#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 2>&1 | tail -3
WLS="cfg5 cfg3 cfg2" tools/gpu_libab.sh ab_k3 build/lib_head.so - > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/r3_summary.py gpurun_out/ab_k3 | grep -v 'synth\|hist_update\|k1t_demod\|k_done\|k4_r900\|dense'
