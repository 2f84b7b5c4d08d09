#!/bin/bash
# A/B of two builds of the library on one box: $1 = path of the other libamrdemod.so; workloads in $W; alternating runs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
ALT=$GRAFT_REPO_ROOT/$1
for w in ${W:-cfg5 cfg3}; do
  for i in 1 2 3; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 > $O/ab_new_${w}_$i.json 2>/dev/null
    AMR_LIB_OVERRIDE=$ALT timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 > $O/ab_old_${w}_$i.json 2>/dev/null
  done
done
W1=$(echo ${W:-cfg5} | cut -d' ' -f1)
bash tools/gpu_timeline.sh ab_new --workload $W1 | grep -v rocclr | sed -n 1,14p
AMR_LIB_OVERRIDE=$ALT bash tools/gpu_timeline.sh ab_old --workload $W1 | grep -v rocclr | sed -n 1,14p
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/ab_*_?.json')):
    try:
        j=json.loads(open(f).read().strip().split('\n')[-1]); r=j['roofline']
        print(f"{f.split('/')[-1]:28s}", j['value'], j['ms_per_step'], j['steady_ms_per_step'], 'k1',r['k1_ms'],'frac',r['frac'],'k2',r['search_ms'],'wp',r['whole_path_frac'],r['whole_path_frac_timed'])
    except Exception as e: print(f, e)
PY
