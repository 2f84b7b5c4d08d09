#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['steady_ms_per_step'], j['pipeline_fill_ms'], j['roofline']['k1_ms'], j['roofline']['frac'], j['roofline']['search_ms'], j['latency_us_per_block'] if 'latency_us_per_block' in j else '')"; done
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null; python -c "
import json;d=json.load(open('$O/bench_driver_cmd.json'));print(d['value'],d['ms_per_step'],d['steady_ms_per_step'],d['pipeline_fill_ms'],d['roofline']['k1_ms'],d['roofline']['frac'],d['roofline']['search_ms'],d['latency_us_per_block'])"
