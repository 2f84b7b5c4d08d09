#!/bin/bash
# round 6: the driver's 20-step line: where the fill goes (timing events on/off, steps 20/40)
cd $GRAFT_REPO_ROOT; O=gpurun_out/fill; mkdir -p $O; : > $O/ab.txt
for rep in 1 2 3; do for v in "20 4 2" "20 4 1" "20 0 1" "20 5 1"; do set -- $v
  timeout 300 python bench.py --steps $1 --warmup 5 --k1-events $2 --k1-level $3 --no-cpu-baseline --no-measure-traffic > $O/b_$1_$2_$3_$rep.json 2> $O/b_$1_$2_$3_$rep.err
  python - $O/b_$1_$2_$3_$rep.json $1 "$2 level $3" >> $O/ab.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"steps {sys.argv[2]} k1-events {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} fill {d['pipeline_fill_ms']} k1_ms {r['k1_ms']}")
PY
done; done; sort $O/ab.txt
