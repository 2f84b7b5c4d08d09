#!/bin/bash
# round 6: chip 8, the search inside the K1 wave (AMR_INWAVE=1, default) against the early search next to K1 (AMR_INWAVE=0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/inwave; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robust.py tests/test_gpu_deferral.py tests/test_gpu_fullsize.py -x -q -m gpu -k "8 or chip8 or round6" 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
for rep in 1 2 3; do for m in 0 1; do
  AMR_INWAVE=$m timeout 300 python bench.py --workload cfg4:8 --steps 100 --warmup 5 --k1-level 2 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${m}_$rep.json 2> $O/b_${m}_$rep.err
  python - $O/b_${m}_$rep.json $m >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"cfg4:8 inwave {sys.argv[2]}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}", d["config"]["checks"].get("last_timed_step","")[:60])
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json','.err')).read()[-400:])
PY
done; done; sort $O/ab.txt
