#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
AMR_K2_DBG=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "k2 phases"
for i in 1 2; do
for cfg in "AMR_X=1" "AMR_K2_IMPL=old"; do
env $cfg timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$cfg', j['value'], j['ms_per_step'], j['roofline']['k1_ms'], j['roofline']['search_ms'], j['config']['hits_per_step_rank0'])"
done; done
