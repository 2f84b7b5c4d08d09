#!/bin/bash
# round 6: cfg5 / cfg3 tail: K3 loads only the half-window that holds hits (product) against the library before it (hs32), and the
# multi-preamble walk limited to two workgroups per CU by LDS (AMR_K2W_LDS_KB=54)
cd $GRAFT_REPO_ROOT; O=gpurun_out/cfg5_ab; mkdir -p $O; : > $O/ab.txt
run() { # name lib env workload
  t=$(echo $4 | tr : _)
  env $3 AMR_LIB_OVERRIDE=$2 timeout 300 python bench.py --workload $4 --steps 60 --warmup 5 --no-cpu-baseline --no-measure-traffic --device-state off > $O/b_${t}_$1_$5.json 2> $O/b_${t}_$1_$5.err
  python - $O/b_${t}_$1_$5.json $4 $1 >> $O/ab.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:8} {sys.argv[3]:>8}: value {d['value']:.0f} ms/step {d['ms_per_step']} steady {d['steady_ms_per_step']} k1_ms {r['k1_ms']} frac {r['frac']} search_ms {r['search_ms']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
P=$GRAFT_REPO_ROOT/rtlamr_amd/csrc/libamrdemod.so; B=$GRAFT_REPO_ROOT/build/libamrdemod_hs32.so
for rep in 1 2; do
  run before $B AMR_X=1 cfg5 $rep; run k3half $P AMR_X=1 cfg5 $rep; run lds54 $P AMR_K2W_LDS_KB=54 cfg5 $rep; run lds40 $P AMR_K2W_LDS_KB=40 cfg5 $rep
  run before $B AMR_X=1 cfg3 $rep; run k3half $P AMR_X=1 cfg3 $rep
done
sort $O/ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/$O/tr
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/tr -o prof --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --depth 1 --steps 10 --warmup 3 --k1-events 0 --no-cpu-baseline --spinup-ms 50 --no-measure-traffic --device-state off > /dev/null 2>&1
S=$(find $GRAFT_REPO_ROOT/$O/tr -name '*kernel_stats.csv' | head -1); cp $S $GRAFT_REPO_ROOT/$O/kernel_stats_cfg5_depth1.csv; rm -rf $GRAFT_REPO_ROOT/$O/tr
head -5 $GRAFT_REPO_ROOT/$O/kernel_stats_cfg5_depth1.csv | cut -c1-140
