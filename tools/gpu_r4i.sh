#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
KINDS="idm all" MODES="warm" bash tools/gpu_k2b.sh k2b_old k2b_ab | grep "k2b \|wave life\|cycles\|##"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py tests/test_gpu_impl_fallbacks.py tests/test_gpu_random.py tests/test_gpu_r900.py tests/test_gpu_validate.py -m gpu -x -q > $O/pytest_i.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_i.log | tail -2
for w in cfg3 cfg5; do for i in 1 2; do timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 > $O/bench_${w}_ab_$i.json 2>/dev/null; echo "$w rc=$?"; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_cfg?_ab_?.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'k2',r['search_ms'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
PY
