#!/bin/bash
# round 6 A/B: the tail behind the end of a multi-round K1 launch (AMR_GATE_END=-1, default) against the gate kernel (0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/gate_end; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do for c in ${CHIPS:-40 32 48}; do for m in 0 -1; do
  AMR_GATE_END=$m timeout 300 python bench.py --workload cfg4:$c --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${c}_${m}_$rep.json 2> $O/b_${c}_${m}_$rep.err
  python - $O/b_${c}_${m}_$rep.json $c $m >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"chip {sys.argv[2]} gate_end {sys.argv[3]:>2}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}")
except Exception as e:
    print("chip", sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done
cat $O/ab.txt
