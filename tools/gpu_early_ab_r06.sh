#!/bin/bash
# round 6: chip 8 with the halo-shift K1: early search on (default at BlockSize <= 512) / off
cd $GRAFT_REPO_ROOT; O=gpurun_out/early_r06; mkdir -p $O; : > $O/ab.txt
for rep in 1 2 3; do for e in 0 1; do
  AMR_EARLY_SEARCH=$e timeout 300 python bench.py --workload cfg4:8 --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${e}_$rep.json 2> $O/b_${e}_$rep.err
  python - $O/b_${e}_$rep.json $e >> $O/ab.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"cfg4:8 early {sys.argv[2]}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}")
PY
done; done; cat $O/ab.txt
