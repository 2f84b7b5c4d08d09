#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b.log; : > $O
run() { echo "## $*" >> $O; timeout 200 "$@" >> $O 2>&1; }
for i in 1 2 3; do
run build/k1b_hk0 s0p1x1n16w1a1l11r13 131072 30 0 1
run build/k1b_hk1 s0p1x1n16w1a1l11r13 131072 30 0 1
done
grep "^##\|^k1b" $O | cut -c1-170
