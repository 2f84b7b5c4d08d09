#!/bin/bash
# K2 harness on the GPU box: every binary given (build/<name>), each with the kinds in $KINDS (default scm idm all), modes in $MODES
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/k2b.log; : > $O
for b in "$@"; do for k in ${KINDS:-scm idm all}; do for m in ${MODES:-warm}; do echo "## $b $k $m" >> $O; timeout 120 build/$b $k ${TILES:-} ${REPS:-} $( [ $m = warm ] || echo $m ) >> $O 2>&1; done; done; done
cat $O
