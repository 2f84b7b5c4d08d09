#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
build/chain_bench 2>&1 | tee $O/chain_bench.txt
for ev in 0 4 8 16 0 4 8 16; do timeout 300 python bench.py --no-cpu-baseline --k1-events $ev 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('k1-events $ev', j['value'], j['ms_per_step'], j['steady_ms_per_step'], j['roofline']['k1_ms'], j['roofline']['frac'], j['roofline']['whole_path_frac'], j['roofline']['whole_path_frac_timed'])"; done
for ev in 0 4 8; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --k1-events $ev 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('20 steps k1-events $ev', j['value'], j['ms_per_step'], j['steady_ms_per_step'], j['roofline']['k1_ms'], j['roofline']['frac'])"; done
