#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
bash tools/gpu_timeline.sh noquery | grep -v rocclr | sed -n 1,16p
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_noquery_$i.json 2>/dev/null; echo "rc=$?"; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferral.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_noquery_?.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'k2',r['search_ms'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
PY
