#!/bin/bash
# how long the gate kernel stays after the K1 launch's last workgroup has started (AMR_GATE_DELAY_TICKS, 100 MHz): 6 us (the
# product) against 1 us and 0, alternating runs on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/gate_delay; mkdir -p $O
for i in 1 2 3; do for D in 600 100 0; do
  AMR_GATE_DELAY_TICKS=$D timeout 300 python bench.py --workload ${W:-cfg2} --no-cpu-baseline --no-verify --steps 200 2>/dev/null | tail -1 > $O/d${D}_$i.json
  python -c "
import json
d=json.loads(open('$O/d${D}_$i.json').read()); r=d['roofline']; print('delay $D', d['value'], d['ms_per_step'], d['steady_ms_per_step'], r['k1_ms'], r['search_ms'])"
done; done
