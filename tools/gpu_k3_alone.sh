#!/bin/bash
# K3 (and K2) on their own: rocprofv3 kernel stats of bench.py --depth 1 (the kernels of a batch one after the other), product
# against variant libraries: tools/gpu_k3_alone.sh "prod walk" "cfg5 cfg2 cfg3"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/k3alone; mkdir -p $O
for W in ${2:-cfg5 cfg2}; do for V in ${1:-prod}; do
  L=$R/rtlamr_amd/csrc/libamrdemod.so; [ $V != prod ] && L=$R/build/libamrdemod_$V.so
  rm -rf $O/p_$V_$W
  AMR_LIB_OVERRIDE=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_${V}_$W -o prof --output-format csv -- python $R/bench.py --workload $W --steps 20 --warmup 5 --k1-events 0 --no-cpu-baseline --depth 1 > $O/${V}_$W.log 2>&1
  S=$(find $O/p_${V}_$W -name '*kernel_stats.csv' | head -1)
  echo "== $V $W: $(grep '^{' $O/${V}_$W.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], str(d['config']['checks']['last_timed_step'])[-28:])")"
  grep -E "k3_slice|k2_search|k5_|k4_" $S | python -c "
import csv,sys
for r in csv.reader(sys.stdin): print(\"   %-56s calls %5s avg %8.1f us\" % (r[0][:56], r[1], float(r[3])/1000))"
  rm -rf $O/p_${V}_$W
done; done
