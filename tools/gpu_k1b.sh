#!/bin/bash
# K1 harness runs on the GPU box: every binary given as argument, all its variants round-robin on the same buffers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b.log; : > $O
for b in "$@"; do echo "## $b" >> $O; timeout 300 build/$b all 131072 40 0 1 >> $O 2>&1; done
cat $O | cut -c1-200
