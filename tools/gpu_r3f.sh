#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3f}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline"
timeout 300 $B --steps 20 --warmup 5 > $O/bench_line.log 2>&1
timeout 300 $B > $O/bench_200.log 2>&1
timeout 300 $B --workload cfg3 --steps 50 > $O/bench_cfg3.log 2>&1
timeout 300 $B --workload cfg5 --steps 50 > $O/bench_cfg5.log 2>&1
timeout 300 $B --workload cfg4:8 --steps 50 > $O/bench_cfg4_8.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in cfg2 cfg3 cfg5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${w} -o prof --output-format csv -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 100 > $R/$O/prof_${w}.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_iso_${w} -o prof --output-format csv -- python $R/bench.py --workload $w --depth 1 --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 100 > $R/$O/prof_iso_${w}.log 2>&1
done
