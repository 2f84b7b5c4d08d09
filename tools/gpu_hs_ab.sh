#!/bin/bash
# round 6: the halo-shift K1 in the pipelined bench: chip 8 (product) and chip 32 (variant hs32) against the library without it (hs0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/hs_ab; mkdir -p $O; : > $O/ab.txt
for rep in 1 2 3; do for v in hs0 hs32; do for w in cfg4:8 cfg4:32; do
  t=$(echo $w | tr : _)
  AMR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/libamrdemod_$v.so timeout 300 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${t}_${v}_$rep.json 2> $O/b_${t}_${v}_$rep.err
  python - $O/b_${t}_${v}_$rep.json $w $v >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:8} {sys.argv[3]:>5}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']} {r['kernel'][:40]}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done; sort $O/ab.txt
