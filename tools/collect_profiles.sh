#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of the default bench command.
# Raw output under gpurun_out/prof_<tag>/; tools/summarise_profiles.py turns it into profiles/<tag>/.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 2 --k1-events 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o prof --output-format csv -- $B > $O/stats.log 2>&1
# counters in their own runs (kernel-trace only), FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/pmc_sq_a -o pmc --output-format csv -- $B > $O/pmc_sq_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/pmc_sq_b -o pmc --output-format csv -- $B > $O/pmc_sq_b.log 2>&1
$B > $O/bench_unprofiled.log 2>&1
python $R/bench.py > $O/bench_line.log 2>&1
