#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b.log; : > $O
run() { echo "## $*" >> $O; timeout 200 "$@" >> $O 2>&1; }
run build/k1b_diag all 131072 15 0 2
run build/k1b_diag old,s0p1x1n16w1a1l11 131072 15 1 1
cat $O
