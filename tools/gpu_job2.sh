#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_r900.py tests/test_gpu_random.py tests/test_gpu_validate.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k4 -o p --output-format csv -- python $R/tests/r900_rate.py > /tmp/prof_k4.log 2>&1; grep "GPU:\|CPU" /tmp/prof_k4.log
python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/prof_k4/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if any(k in n for k in ('k1','k2','k3','k4','k_')): print('   %-70s calls %s avg %.1f us min %.1f' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
