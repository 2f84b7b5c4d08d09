#!/bin/bash
# round 4, first K2-row measurement: harness warm + cold, parity tests that cover the search, bench cfg2 depth 3 / 1
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
{ echo "## warm"; timeout 120 build/k2b scm; echo "## cold"; timeout 120 build/k2b scm 2049 40 cold; } > $O/k2b_row.log 2>&1
cat $O/k2b_row.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py tests/test_gpu_deferral.py tests/test_gpu_comm.py tests/test_gpu_impl_fallbacks.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log; tail -5 $O/pytest_a.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_row_d3.json 2> $O/bench_row_d3.err; echo "bench rc=$?"; tail -c 900 $O/bench_row_d3.json
timeout 300 python bench.py --no-cpu-baseline --depth 1 > $O/bench_row_d1.json 2> $O/bench_row_d1.err; echo "bench d1 rc=$?"; tail -c 600 $O/bench_row_d1.json
timeout 300 python bench.py --gpus 2 ; echo "bench --gpus 2 rc=$?"
