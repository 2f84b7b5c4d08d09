#!/bin/bash
# K1 harness at other block sizes: tools/gpu_k1b_bs.sh <binary> <BlockSize> [blocks]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b_bs$2.log; : > $O
B=${3:-$((2147483648 / ($2 * 2)))}
for rep in 1 2; do timeout 300 build/$1 all $B 30 0 1 $2 >> $O 2>&1; done
cat $O | cut -c1-220
