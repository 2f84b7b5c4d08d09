#!/bin/bash
# round 6 A/B: the tail's gate as a wait packet (hipStreamWaitValue64, default) against the gate kernel (AMR_GATE_MODE=kernel)
cd $GRAFT_REPO_ROOT; O=gpurun_out/gate_mode; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do for w in ${WL:-cfg2 cfg4:40 cfg4:32 cfg4:48 cfg3 cfg5 cfg4:8}; do for m in kernel value; do
  t=$(echo $w | tr : _)
  AMR_GATE_MODE=$m timeout 300 python bench.py --workload $w --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${t}_${m}_$rep.json 2> $O/b_${t}_${m}_$rep.err
  python - $O/b_${t}_${m}_$rep.json $w $m >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:8} gate {sys.argv[3]:>6}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done
cat $O/ab.txt
