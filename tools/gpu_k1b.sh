#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b.log; : > $O
run() { echo "## $*" >> $O; timeout 200 "$@" >> $O 2>&1; }
run build/k1b_sw all 131072 40 0 1
grep "^k1b" $O | cut -c1-150
