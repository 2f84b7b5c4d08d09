#!/bin/bash
# Why is the K1 round behind the two-lanes-per-row search slow (VERDICT r04 #2a)?  Workgroup timelines of K1 inside the
# pipelined bench (cfg3: two K1 rounds per batch; cfg2: one), product search (walk / row) against -DAMR_K2R_LPR2=1, both built
# with -DAMR_K1T_CLK=1 (tools/build_variant.sh tl "-DAMR_K1T_CLK=1"; tools/build_variant.sh tl_lpr2 "-DAMR_K1T_CLK=1 -DAMR_K2R_LPR2=1"),
# each with the gate kernel where round 4 had it (AMR_GATE_EVENT=0) and behind the event in front of the K1 launch it waits for.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/k1tl; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'k1', r['k1_ms'], 'k2', r['search_ms'])"; }
run() {  # variant workload launches-per-batch gate-event extra-env
  local T=$1_$(echo $2 | tr ':' '_')_g$4$5
  env AMR_GATE_EVENT=$4 $6 AMR_K1_TIMELINE=$GRAFT_REPO_ROOT/$O/raw_$T.txt AMR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/libamrdemod_$1.so timeout 300 python bench.py --workload $2 --no-cpu-baseline --no-verify --steps 40 --k1-events 0 > $O/$T.log 2>&1
  line $O/$T.log
  python tools/k1_timeline_report.py $O/raw_$T.txt $3 > $O/report_$T.txt 2>&1; grep "^# launch [0-9]" $O/report_$T.txt | cut -c1-220; rm -f $O/raw_$T.txt
}
for G in 0 1; do
  run tl cfg3 2 $G
  run tl_lpr2 cfg3 2 $G
  run tl cfg2 1 $G
done
