#!/bin/bash
# round 6: cfg3 (idm), K3 limited in workgroups per CU (AMR_K3_LDS_KB) / at wave priority, while it shares the chip with the search
cd $GRAFT_REPO_ROOT; O=gpurun_out/cfg3_k3lds; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do for v in "0 0" "24 0" "30 0" "40 0" "0 3"; do set -- $v
  AMR_K3_LDS_KB=$1 AMR_K3_PRIO=$2 timeout 300 python bench.py --workload cfg3 --steps 40 --warmup 5 --k1-level 2 --no-cpu-baseline --no-measure-traffic --device-state off --spinup-ms 100 > $O/b_$1_$2_$rep.json 2> $O/b_$1_$2_$rep.err
  python - $O/b_$1_$2_$rep.json $1 $2 >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"k3lds {sys.argv[2]:>3} prio {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']} k1_ms {r['k1_ms']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; sort $O/ab.txt
