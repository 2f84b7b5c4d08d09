#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
K2B_BURSTS=2 KINDS=scm MODES="warm" TILES=2049 REPS=40 bash tools/gpu_k2b.sh k2b | grep "k2b scm\|k2r scm\|wave life\|cycles\|##\|tile"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py tests/test_gpu_impl_fallbacks.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f.log; tail -3 $O/pytest_f.log
bash tools/gpu_timeline.sh taps21 | grep -v rocclr | sed -n 1,12p
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_taps21_$i.json 2>/dev/null; echo "rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_taps21_?.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
PY
