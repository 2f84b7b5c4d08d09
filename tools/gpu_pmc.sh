#!/bin/bash
# PMC passes (kernel-trace only) of one workload at --depth 1: $1 = tag, $2 = workload
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc}; rm -rf $O; mkdir -p $O
W=${2:-cfg3}
B="python $R/bench.py --workload $W --depth ${3:-1} --steps 6 --warmup 2 --no-cpu-baseline --spinup-ms 30"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o pmc --output-format csv -- $B > $O/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o pmc --output-format csv -- $B > $O/write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/sqa -o pmc --output-format csv -- $B > $O/sqa.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAVES -d $O/sqb -o pmc --output-format csv -- $B > $O/sqb.log 2>&1
ls $O/*/
