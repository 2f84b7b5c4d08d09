#!/bin/bash
# round 6 (second half): every chip length of cfg4 + cfg2 / cfg3 / cfg5 on the current library, 100 steps each; AMR_GATE_DELAY_TICKS A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-sweep_b}; mkdir -p $O; : > $O/sweep.txt
line() { python - "$@" >> $O/sweep.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:8} {sys.argv[3]:>10}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} whole {r['whole_path_frac']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
for w in cfg2 cfg4:8 cfg4:32 cfg4:40 cfg4:48 cfg4:56 cfg4:64 cfg4:72 cfg4:80 cfg4:88 cfg4:96 cfg3 cfg5; do
  t=$(echo $w | tr : _)
  timeout 300 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_$t.json 2> $O/b_$t.err
  line $O/b_$t.json $w default
done
for rep in 1 2; do for w in cfg2 cfg4:40 cfg3; do for dly in 0 200 600; do
  t=$(echo $w | tr : _)
  AMR_GATE_DELAY_TICKS=$dly timeout 300 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/d_${t}_${dly}_$rep.json 2> $O/d_${t}_${dly}_$rep.err
  line $O/d_${t}_${dly}_$rep.json $w delay$dly
done; done; done
cat $O/sweep.txt
