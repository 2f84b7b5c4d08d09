#!/bin/bash
# K1 harness, ring-size experiment (round 6): build/k1b_c<CL>r<RING> binaries, BlockSize 2048 (chip 32..48) / 512 (chip 8), 1 GiB
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1b_ring.log; : > $O
for rep in 1 2; do
for b in k1b_c40r0 k1b_c40r56 k1b_c40r64 k1b_c48r0 k1b_c48r64 k1b_c32r0 k1b_c32r64; do echo "## $b" >> $O; timeout 120 build/$b all 262144 30 0 1 2048 >> $O 2>&1; done
for b in k1b_c8r0 k1b_c8r64; do echo "## $b" >> $O; timeout 120 build/$b all 1048576 30 0 1 512 >> $O 2>&1; done
done
cut -c1-220 $O
