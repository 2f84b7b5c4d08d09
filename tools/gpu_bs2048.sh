#!/bin/bash
# round 6: why K1 is 7-17 % slower in the pipelined bench than alone at BlockSize 2048 (chip 32 / 40 / 48)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bs2048; mkdir -p $O
for c in 40 32; do
  rm -rf $O/tl_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tl_$c -o prof --output-format csv -- python $R/bench.py --workload cfg4:$c --steps 20 --warmup 5 --k1-events 0 --no-cpu-baseline --spinup-ms 150 --no-measure-traffic --device-state off > $O/trace_$c.log 2>&1
  F=$(find $O/tl_$c -name '*kernel_trace.csv' | head -1)
  python $R/tools/timeline.py $F 4 > $O/timeline_$c.txt 2>&1
  S=$(find $O/tl_$c -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats_$c.csv
  rm -rf $O/tl_$c
done
cd $R
for c in 40 32 72; do
  AMR_LIB_OVERRIDE=$R/build/libamrdemod_tl4k.so AMR_K1_TIMELINE=$O/k1tl_$c.txt timeout 300 python bench.py --workload cfg4:$c --steps 20 --warmup 5 --no-cpu-baseline --spinup-ms 150 --no-measure-traffic --device-state off > $O/bench_tl_$c.json 2> $O/bench_tl_$c.err
  python tools/k1_timeline_report.py $O/k1tl_$c.txt > $O/k1tl_report_$c.txt 2>&1
  rm -f $O/k1tl_$c.txt
done
timeout 120 build/k1b_c40tl all 262144 20 0 1 2048 > $O/k1b_c40tl.txt 2>&1
tail -12 $O/timeline_40.txt; tail -30 $O/k1tl_report_40.txt | cut -c1-230; cut -c1-250 $O/k1b_c40tl.txt
