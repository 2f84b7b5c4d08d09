#!/bin/bash
# gated early tail vs round 3's host-launched tail: timelines + bench lines on one box; parity subset first
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferral.py tests/test_gpu_comm.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_c.log 2>&1; echo "pytest rc=$?" >> $O/pytest_c.log; tail -3 $O/pytest_c.log
bash tools/gpu_timeline.sh gate | grep -v rocclr | head -40
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $O/bench_gate_$i.json 2>/dev/null; echo "gate rc=$?"
AMR_HOST_TAIL=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_hosttail_$i.json 2>/dev/null; echo "host rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_gate_*.json')+glob.glob('gpurun_out/r04/bench_hosttail_*.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f.split('/')[-1], j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'search',r['search_ms'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'], [v[:40] for v in j['config']['checks'].values()][:1])
PY
