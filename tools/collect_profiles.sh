#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of the default bench command, then the
# unprofiled bench lines of every workload.  Raw output under gpurun_out/prof_<tag>/; tools/summarise_profiles.py turns
# it into profiles/<tag>/.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --k1-events 1 --no-cpu-baseline --spinup-ms 150"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o prof --output-format csv -- $B > $O/stats.log 2>&1
# counters in their own runs (kernel-trace only), FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/pmc_sq_a -o pmc --output-format csv -- $B > $O/pmc_sq_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/pmc_sq_b -o pmc --output-format csv -- $B > $O/pmc_sq_b.log 2>&1
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1                       # the driver's command line
python bench.py --no-cpu-baseline > $O/bench_200steps.log 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 50 > $O/bench_cfg3.log 2>&1
python bench.py --workload cfg5 --no-cpu-baseline --steps 50 > $O/bench_cfg5.log 2>&1
for c in 8 32 40 48 56 64 80 88 96; do python bench.py --workload cfg4:$c --no-cpu-baseline --steps 50 > $O/bench_cfg4_$c.log 2>&1; done
python bench.py --validate --no-cpu-baseline > $O/bench_validate.log 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-260; done
