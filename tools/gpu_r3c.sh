#!/bin/bash
# round 3, GPU call B: suite + benches + kernel stats after the walk kernel's unconditional loads and the TU split
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1
timeout 300 $B > $O/bench_200.log 2>&1
for w in cfg3 cfg5; do timeout 300 $B --workload $w --steps 50 > $O/bench_$w.log 2>&1; done
for c in 8 32 40 48 56 64 80 88 96; do timeout 300 $B --workload cfg4:$c --steps 50 > $O/bench_cfg4_$c.log 2>&1; done
timeout 300 $B --blocks 100000 --steps 50 > $O/bench_100000.log 2>&1
timeout 300 $B --validate > $O/bench_validate.log 2>&1
AMR_BENCH_FORCE_DIST=1 timeout 300 $B --steps 20 --warmup 5 > $O/bench_dist1.log 2>&1
timeout 300 $B --steps 20 --warmup 5 --spinup-ms 0 > $O/bench_nospin.log 2>&1
timeout 120 python tools/single_block_rate.py > $O/single_block.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in cfg2 cfg3 cfg5 cfg4:8; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${w/:/_} -o prof --output-format csv -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 100 > $R/$O/prof_${w/:/_}.log 2>&1
done
cd $R
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-330; done > $O/summary.txt
tail -4 $O/pytest_gpu.log >> $O/summary.txt; tail -3 $O/single_block.log >> $O/summary.txt
cat $O/summary.txt
