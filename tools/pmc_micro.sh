#!/bin/bash
# PMC passes over the DMA/store microbenchmark (developer diagnostics); small passes, tight timeouts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_micro; rm -rf $O; mkdir -p $O
i=0
while read -r line; do
  i=$((i+1))
  timeout 25 rocprofv3 --kernel-trace --pmc $line -d $O/p$i -o pmc --output-format csv -- $R/build/dma_bench2 short > $O/p$i.log 2>&1
done <<'LIST'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
LIST
