#!/bin/bash
# one-call check of a library change: K2 harness (kinds in $KINDS), the parity tests that cover the search, then the A/B
# of tools/gpu_libab2.sh against the library given as $1 (workloads in $W)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
KINDS="${KINDS:-idm}" MODES="warm" bash tools/gpu_k2b.sh k2b | grep "k2b \|k2r \|wave life\|##\|tile "
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py tests/test_gpu_impl_fallbacks.py tests/test_gpu_random.py tests/test_gpu_r900.py tests/test_gpu_validate.py tests/test_gpu_deferral.py -m gpu -x -q > $O/pytest_ab.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_ab.log | tail -2
rm -f $O/ab_*; bash tools/gpu_libab2.sh $1 | tail -14
