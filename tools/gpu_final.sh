#!/bin/bash
# end-of-round run on one box: the whole GPU suite, the soak, then the profile collection
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final_tests.log 2>&1; echo "suite exit $?" >> gpurun_out/final_tests.log
tools/gpu_soak.sh ${1:-3000} 6 > /dev/null 2>&1
tools/collect_profiles_r03.sh r03 > gpurun_out/collect_r03.log 2>&1
tail -3 gpurun_out/final_tests.log; tail -4 gpurun_out/soak_r03.txt; tail -40 gpurun_out/collect_r03.log
