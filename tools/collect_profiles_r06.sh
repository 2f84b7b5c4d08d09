#!/bin/bash
# Round 6, on the GPU box (via gpurun): rocprofv3 kernel stats (pipelined and --depth 1) of cfg2 / cfg3 / cfg5, the FETCH_SIZE /
# WRITE_SIZE passes of cfg2 (K1's traffic) and of cfg5 (K2 / K3 re-reads, VERDICT r05 #2c), the kernel timeline of cfg2, then
# unprofiled bench lines: the driver's command (traffic measured in the run, device state from sysfs), 200 steps, cfg3,
# cfg5, validated, six fresh processes.  Raw output under gpurun_out/prof_<tag>/; tools/summarise_profiles.py <tag> turns it
# into profiles/<tag>/.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
Q="--no-cpu-baseline --device-state off"
for w in cfg2 cfg3 cfg5; do
  B="python $R/bench.py --workload $w --steps 20 --warmup 5 --k1-events 1 $Q --spinup-ms 150"
  D="$B --depth 1"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$w/stats -o prof --output-format csv -- $B > $O/$w.stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$w/stats_iso -o prof --output-format csv -- $D > $O/$w.stats_iso.log 2>&1
  # counters in their own runs (kernel-trace only); FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots)
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$w/pmc_fetch -o pmc --output-format csv -- $D > $O/$w.pmc_fetch.log 2>&1
  if [ $w != cfg3 ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$w/pmc_write -o pmc --output-format csv -- $D > $O/$w.pmc_write.log 2>&1
  fi
done
cd $R
F=$(find $O/cfg2/stats -name '*kernel_trace.csv' | head -1); python tools/timeline.py $F 6 > $O/timeline_cfg2.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_line.log 2>&1                    # the driver's command line (measures K1's traffic itself)
python bench.py --no-cpu-baseline > $O/bench_200steps.log 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 50 > $O/bench_cfg3.log 2>&1
python bench.py --workload cfg5 --no-cpu-baseline --steps 50 > $O/bench_cfg5.log 2>&1
python bench.py --validate --no-cpu-baseline > $O/bench_validate.log 2>&1
AMR_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_1rank.log 2>&1
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); d=j['device']['before_timed_region'] or {}; print(j['value'], j['ms_per_step'], j['steady_ms_per_step'], j['pipeline_fill_ms'], j['roofline']['k1_ms'], j['roofline']['frac'], j['roofline']['whole_path_frac'], d.get('sclk_mhz'), d.get('power_w'))"; done > $O/fresh_runs.txt 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-220; done
cat $O/fresh_runs.txt
