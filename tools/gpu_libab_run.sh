#!/bin/bash
WLS="cfg2 cfg3" tools/gpu_libab.sh ab_pf build/lib_head.so - build/lib_pf8.so build/lib_pf11.so > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/r3_summary.py gpurun_out/ab_pf | grep -v 'synth\|hist_update\|k1t_demod'
