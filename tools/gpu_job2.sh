#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('new', j['value'], j['ms_per_step'], j['roofline']['k1_ms'], j['roofline']['search_ms'], j['config']['hits_per_step_rank0'])"
AMR_K1_IMPL=old timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('old', j['value'], j['ms_per_step'], j['roofline']['k1_ms'], j['roofline']['search_ms'], j['config']['hits_per_step_rank0'])"
done
