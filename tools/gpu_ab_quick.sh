#!/bin/bash
# A/B of the in-tree library against another build (AMR_LIB_OVERRIDE) on one box: alternating bench runs, plain and validated,
# then the standalone kernel durations (--depth 1) of both.  usage: gpu_ab_quick.sh <other libamrdemod.so> [workload]
cd $GRAFT_REPO_ROOT; OTHER=$1; W=${2:-cfg2}
run() { python bench.py --workload $W --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "tree   plain    $(run)";            echo "other  plain    $(AMR_LIB_OVERRIDE=$OTHER run)"
  echo "tree   validate $(run --validate)"; echo "other  validate $(AMR_LIB_OVERRIDE=$OTHER run --validate)"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for lib in tree other; do for mode in "" --validate; do
rm -rf $O/d1; [ $lib = other ] && export AMR_LIB_OVERRIDE=$OTHER || unset AMR_LIB_OVERRIDE
timeout 300 rocprofv3 --kernel-trace --stats -d $O/d1 -o prof --output-format csv -- python $R/bench.py --workload $W --steps 30 --warmup 5 --k1-events 0 --no-cpu-baseline --depth 1 $mode > $O/d1.log 2>&1
S=$(find $O/d1 -name '*kernel_stats.csv' | head -1)
echo "== $lib ${mode:-plain}"; python - $S <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("k3_slice","k5_","k2_search")) and int(r["Calls"])>10:
        print(f'  {n[:56]:56s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done; done; rm -rf $O/d1
