#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
AMR_GATE_LATE=1 bash tools/gpu_timeline.sh gatelate | grep -v rocclr | sed -n 14,30p
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; echo "$tag rc=$?"; }
for i in 1 2; do
run gate_$i A=1
run gatelate_$i AMR_GATE_LATE=1
run gated0_$i AMR_GATE_DELAY=0
run gated2000_$i AMR_GATE_DELAY=2000
run hosttail_$i AMR_HOST_TAIL=1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_gate*_?.json')+glob.glob('gpurun_out/r04/bench_hosttail_?.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
PY
