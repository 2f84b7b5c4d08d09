cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/k1tl; mkdir -p $O
for C in 80 88; do
AMR_K1_TIMELINE=$GRAFT_REPO_ROOT/$O/raw_c$C.txt AMR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/libamrdemod_tl_c.so timeout 300 python bench.py --workload cfg4:$C --no-cpu-baseline --no-verify --steps 60 --k1-events 0 > $O/tl_c$C.log 2>&1
python tools/k1_timeline_report.py $O/raw_c$C.txt 1 > $O/report_tl_c$C.txt 2>&1; tail -7 $O/report_tl_c$C.txt | cut -c1-230
AMR_K1_TIMELINE=$GRAFT_REPO_ROOT/$O/raw_c${C}_d1.txt AMR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/libamrdemod_tl_c.so timeout 300 python bench.py --workload cfg4:$C --no-cpu-baseline --no-verify --steps 60 --k1-events 0 --depth 1 > $O/tl_c${C}_d1.log 2>&1
python tools/k1_timeline_report.py $O/raw_c${C}_d1.txt 1 > $O/report_tl_c${C}_d1.txt 2>&1; tail -4 $O/report_tl_c${C}_d1.txt | cut -c1-230
done
