#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/dbg2.log; : > $O
export AMR_RANDOM_SEEDS=3000
for id in "test_random_configuration[1097]" "test_random_configuration[571]" "test_random_pipeline_and_validation[52]" "test_random_sharding[884]"; do
  echo "## $id" >> $O
  timeout 120 python -m pytest "tests/test_gpu_random.py::$id" -q -p no:cacheprovider -x 2>&1 | grep -v '^  File\|Extension modules' | head -30 >> $O
  echo "## $id AMR_DEBUG_SYNC" >> $O
  AMR_DEBUG_SYNC=1 timeout 120 python -m pytest "tests/test_gpu_random.py::$id" -q -p no:cacheprovider -x -s 2>&1 | grep -v '^  File\|Extension modules' | tail -25 >> $O
done
cat $O
