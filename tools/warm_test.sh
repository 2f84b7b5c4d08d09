#!/bin/bash
# does a longer warm-up change which K1 duration mode (0.204 / 0.220 ms) a process lands in?
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for w in 5 400; do
    python bench.py --no-cpu-baseline --warmup $w 2>&1 | grep '^{"metric' | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('warmup', $w, j['value'], j['ms_per_step'], j['roofline']['k1_ms'])"
  done
done
