#!/bin/bash
# round 6: cfg5, K2 walk and K3 limited in workgroups per CU by LDS padding (AMR_K2W_LDS_KB x AMR_K3_LDS_KB)
cd $GRAFT_REPO_ROOT; O=gpurun_out/cfg5_lds; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do for k3 in 0 30 40 54 80; do for k2 in 0 40 54; do
  AMR_K3_LDS_KB=$k3 AMR_K2W_LDS_KB=$k2 timeout 300 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off --spinup-ms 100 > $O/b_${k3}_${k2}_$rep.json 2> $O/b_${k3}_${k2}_$rep.err
  python - $O/b_${k3}_${k2}_$rep.json $k3 $k2 >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"k3lds {sys.argv[2]:>3} k2lds {sys.argv[3]:>3}: value {d['value']:.0f} ms/step {d['ms_per_step']} k1_ms {r['k1_ms']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done; sort $O/ab.txt
