#!/bin/bash
# standalone durations of the tail kernels (bench --depth 1: nothing runs next to them), validated and plain, per workload
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for w in ${WORKLOADS:-cfg2}; do for mode in --validate ""; do
rm -rf $O/d1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/d1 -o prof --output-format csv -- python $R/bench.py --workload $w --steps 30 --warmup 5 --k1-events 0 --no-cpu-baseline --depth 1 $mode > $O/d1.log 2>&1
S=$(find $O/d1 -name '*kernel_stats.csv' | head -1)
echo "== $w ${mode:-plain}"; python - $S <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("k3_slice","k5_","k_done","k2_search","k1t_demod","k1_demod")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}')
PY
done; done
rm -rf $O/d1
