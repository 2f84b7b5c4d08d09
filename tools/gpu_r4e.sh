#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python -c "
import ctypes
hip=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); print('prio range rc', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), 'least', lo.value, 'greatest', hi.value)"
AMR_TAIL_PRIO=-1 bash tools/gpu_timeline.sh tailhi | grep -v rocclr | sed -n 14,24p
AMR_COMPUTE_PRIO=-1 bash tools/gpu_timeline.sh comphi | grep -v rocclr | sed -n 14,24p
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; echo "$tag rc=$?"; }
for i in 1 2; do
run gate_$i A=1
run gatetailhi_$i AMR_TAIL_PRIO=-1
run gatetaillo_$i AMR_TAIL_PRIO=1
run gatecomphi_$i AMR_COMPUTE_PRIO=-1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_gate*_?.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
    print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
PY
