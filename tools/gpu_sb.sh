#!/bin/bash
# the single-block loop (main.go:235's shape): the suite, the rate, the kernel breakdown
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb; rm -rf $O; mkdir -p $O
cd $R; timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 2>&1 | tail -2
python tools/single_block_rate.py 2>&1 | tail -1 | tee $O/single_block.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o prof --output-format csv -- python $R/tools/single_block_rate.py > $O/log.txt 2>&1
cd $R
python - <<'PY'
import csv,glob
for p in glob.glob('gpurun_out/sb/p/**/prof_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(p)))[:6]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} us")
PY
tools/gpu_soak.sh ${1:-3000} 8 > /dev/null 2>&1; tail -3 gpurun_out/soak_r03.txt
