#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o r01 -f csv -- $B > gpurun_out/prof_stats.log 2>&1
B2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-include-regex 'k1_demod|k2_search|k3_slice' -d gpurun_out/pmc_a -o r01 -f csv -- $B2 > gpurun_out/pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex 'k1_demod|k2_search|k3_slice' -d gpurun_out/pmc_b -o r01 -f csv -- $B2 > gpurun_out/pmc_b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex 'k1_demod|k2_search|k3_slice' -d gpurun_out/pmc_c -o r01 -f csv -- $B2 > gpurun_out/pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-include-regex 'k1_demod|k2_search|k3_slice' -d gpurun_out/pmc_d -o r01 -f csv -- $B2 > gpurun_out/pmc_d.log 2>&1
ls -R gpurun_out | head -50
