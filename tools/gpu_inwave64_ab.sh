#!/bin/bash
# round 6: in-wave search at rows of 64 words (chip 32 / 40, AMR_INWAVE=2) against the search launch (AMR_INWAVE=1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/inwave64; mkdir -p $O; : > $O/ab.txt
for c in 32 40; do for m in 1 2; do
  AMR_INWAVE=$m AMR_K1_COOP_MAX=0 timeout 300 python tests/pipelined_probe.py scm $c 256 7 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('probe chip $c inwave $m:', d['n_hits'], d['hits_sha256'][:16], d['pkt_sha256'][:16], d['describe'][-40:])" >> $O/ab.txt 2>&1
done; done
for rep in 1 2 3; do for c in 32 40; do for m in 1 2; do
  AMR_INWAVE=$m timeout 300 python bench.py --workload cfg4:$c --steps 100 --warmup 5 --k1-level 2 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${c}_${m}_$rep.json 2> $O/b_${c}_${m}_$rep.err
  python - $O/b_${c}_${m}_$rep.json $c $m >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"cfg4:{sys.argv[2]} inwave {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}", d["config"]["checks"].get("last_timed_step","")[-40:])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done; done; done; sort $O/ab.txt
