#!/bin/bash
# round 3, GPU call A: the whole GPU suite, then benches of the headline and the wide-row workloads with the walk search
# (default) and the second-generation stream search (AMR_K2_IMPL=stream), deferral, the RCCL path with one rank, kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
AMR_K2_IMPL=stream timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferral.py tests/test_gpu_comm.py -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_stream.log 2>&1
echo "pytest rc=$?" >> $O/pytest_stream.log
B="python bench.py --no-cpu-baseline"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1
AMR_K2_IMPL=stream timeout 300 $B --steps 20 --warmup 5 > $O/bench_line_stream.log 2>&1
timeout 300 $B > $O/bench_200.log 2>&1
AMR_K2_IMPL=stream timeout 300 $B > $O/bench_200_stream.log 2>&1
for w in cfg3 cfg5; do
  timeout 300 $B --workload $w --steps 50 > $O/bench_$w.log 2>&1
  AMR_K2_IMPL=stream timeout 300 $B --workload $w --steps 50 > $O/bench_${w}_stream.log 2>&1
done
for c in 8 32 64 80 96; do timeout 300 $B --workload cfg4:$c --steps 50 > $O/bench_cfg4_$c.log 2>&1; done
timeout 300 $B --blocks 100000 --steps 50 > $O/bench_100000.log 2>&1
timeout 300 $B --blocks 163840 --steps 50 > $O/bench_1p25GiB.log 2>&1
AMR_BENCH_FORCE_DIST=1 timeout 300 $B --steps 20 --warmup 5 > $O/bench_dist1.log 2>&1
AMR_BENCH_FORCE_DIST=1 AMR_BENCH_SHARD=3 timeout 300 $B --steps 20 --warmup 5 --gather raw > $O/bench_dist1_raw_shard3.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in cfg2 cfg3 cfg5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o prof --output-format csv -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 100 > $R/$O/prof_$w.log 2>&1
done
cd $R
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-400; done > $O/summary.txt
tail -5 $O/pytest_gpu.log >> $O/summary.txt
tail -3 $O/pytest_stream.log >> $O/summary.txt
cat $O/summary.txt
