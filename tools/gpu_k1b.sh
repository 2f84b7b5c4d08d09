#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b.log; : > $O
run() { echo "## $*" >> $O; timeout 200 "$@" >> $O 2>&1; }
run build/k1b_tl s0p1x1n16w1a1l11r13 131072 20 0 1 4096
run build/k1b_tl s0p1x1n16w1a1l11r13 65536 20 0 1 8192
run build/k1b_tl s0p1x1n16w1a1l11r13 262144 10 0 1 8192
run build/k1b_tl s0p1x1n16w1a1l11r13 524288 10 0 1 4096
run build/k1b_tl s0p1x1n16w1a1l11r13 262144 20 0 1 2048
cat $O | grep -v "per XCD"
