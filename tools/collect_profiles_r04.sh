#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats (pipelined and --depth 1) of the bench command for cfg2 (the
# driver's command), cfg3 and cfg5, PMC passes (at --depth 1: with counters the profiler runs one kernel at a time, and
# the gate kernel of the pipelined tail would wait out its time-out for a K1 that cannot start beside it), the kernel
# timeline of the pipelined bench, then the unprofiled bench lines of every workload, all on ONE box in one call.
# Raw output under gpurun_out/prof_<tag>/; tools/summarise_profiles.py <tag> turns it into profiles/<tag>/.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg5; do
  B="python $R/bench.py --workload $w --steps 20 --warmup 5 --k1-events 1 --no-cpu-baseline --spinup-ms 150"
  D="$B --depth 1"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$w/stats -o prof --output-format csv -- $B > $O/$w.stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$w/stats_iso -o prof --output-format csv -- $D > $O/$w.stats_iso.log 2>&1
  # counters in their own runs (kernel-trace only); FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots)
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$w/pmc_fetch -o pmc --output-format csv -- $D > $O/$w.pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$w/pmc_write -o pmc --output-format csv -- $D > $O/$w.pmc_write.log 2>&1
  if [ $w = cfg2 ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/$w/pmc_sq_a -o pmc --output-format csv -- $D > $O/$w.pmc_sq_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/$w/pmc_sq_b -o pmc --output-format csv -- $D > $O/$w.pmc_sq_b.log 2>&1
  # the second input distribution: uniform random bytes (LUT bank conflicts)
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/uniform/pmc_sq_b -o pmc --output-format csv -- $D --data uniform > $O/uniform.pmc_sq_b.log 2>&1
  fi
done
cd $R
# kernel timeline of the pipelined bench (start / end of every dispatch of the last steps)
F=$(find $O/cfg2/stats -name '*kernel_trace.csv' | head -1); python tools/timeline.py $F 6 > $O/timeline_cfg2.txt 2>&1
F=$(find $O/cfg5/stats -name '*kernel_trace.csv' | head -1); python tools/timeline.py $F 3 > $O/timeline_cfg5.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1                       # the driver's command line
python bench.py --no-cpu-baseline > $O/bench_200steps.log 2>&1
python bench.py --no-cpu-baseline --data uniform > $O/bench_uniform.log 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 50 > $O/bench_cfg3.log 2>&1
python bench.py --workload cfg5 --no-cpu-baseline --steps 50 > $O/bench_cfg5.log 2>&1
for c in 8 32 40 48 56 64 72 80 88 96; do python bench.py --workload cfg4:$c --no-cpu-baseline --steps 50 > $O/bench_cfg4_$c.log 2>&1; done
python bench.py --validate --no-cpu-baseline > $O/bench_validate.log 2>&1
python bench.py --blocks 100000 --steps 50 --no-cpu-baseline > $O/bench_100000blocks.log 2>&1
python bench.py --blocks 163840 --steps 50 --no-cpu-baseline > $O/bench_1p25GiB.log 2>&1
python bench.py --steps 20 --warmup 5 --spinup-ms 0 --no-cpu-baseline > $O/bench_nospinup.log 2>&1
python bench.py --depth 1 --no-cpu-baseline --steps 50 > $O/bench_depth1.log 2>&1
AMR_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_1rank.log 2>&1
AMR_BENCH_FORCE_DIST=1 AMR_BENCH_SHARD=3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gather raw > $O/bench_rccl_1rank_raw_shard3.log 2>&1
python bench.py --gpus 2 > $O/bench_gpus2_refusal.txt 2>&1; echo "exit code $?" >> $O/bench_gpus2_refusal.txt
python tools/single_block_rate.py > $O/single_block.txt 2>&1
for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['steady_ms_per_step'], j['roofline']['k1_ms'], j['roofline']['frac'], j['roofline']['whole_path_frac'])"; done > $O/fresh_runs.txt 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-200; done
cat $O/fresh_runs.txt; cat $O/bench_gpus2_refusal.txt
