#!/bin/bash
# round 6: wave priority of K3 (AMR_K3_PRIO) while it shares the chip with the next batch's search
cd $GRAFT_REPO_ROOT; O=gpurun_out/k3prio; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do for w in cfg5 cfg3 cfg2; do for p in 0 1 3; do
  t=$(echo $w | tr : _)
  AMR_K3_PRIO=$p timeout 300 python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off --spinup-ms 100 > $O/b_${t}_${p}_$rep.json 2> $O/b_${t}_${p}_$rep.err
  python - $O/b_${t}_${p}_$rep.json $w $p >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:6} prio {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']} k1_ms {r['k1_ms']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done; sort $O/ab.txt
