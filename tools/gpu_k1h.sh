#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k1h.log; : > $O
for rep in 1 2 3; do
for cl in 8 72; do
  bs=4096; nb=131072; if [ $cl = 8 ]; then bs=512; nb=1048576; fi
  for v in h0 h1; do echo "## $v $cl" >> $O; timeout 300 build/k1b_${v}_$cl def $nb 40 0 1 $bs >> $O 2>&1; done
done; done
grep "^k1b\|^##" $O | cut -c1-170 | sed 's/alloc=0(0x[0-9a-f]*)//'
