#!/bin/bash
# the driver's round-end commands + a soak, on one box: pytest -m gpu, smoke(), the bench line, 3000 random seeds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
bash tools/gpu_fullsuite.sh
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('200 steps', j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1', r['k1_ms'], r['frac'], 'k2', r['search_ms'], 'wp', r['whole_path_frac'])"; done
TAG=r04_final bash tools/gpu_soak.sh 3000 6 | tail -4
