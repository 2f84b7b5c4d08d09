#!/bin/bash
# round 6 A/B: the next K1 held back by K3's workgroup counter (AMR_K3_CTR=1, default) or by k_done's word (0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/k3ctr; mkdir -p $O; : > $O/ab.txt
for rep in 1 2 3 4; do for w in ${WL:-cfg2 cfg4:40 cfg3}; do for m in 0 1; do
  t=$(echo $w | tr : _)
  AMR_K3_CTR=$m timeout 300 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${t}_${m}_$rep.json 2> $O/b_${t}_${m}_$rep.err
  python - $O/b_${t}_${m}_$rep.json $w $m >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:8} k3ctr {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done; sort $O/ab.txt
