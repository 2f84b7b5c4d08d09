#!/bin/bash
# the driver's round-end commands on one box: pytest -m gpu, smoke, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/${ROUND:-r06}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log; tail -15 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"; tail -c 1500 $O/bench_driver_cmd.json
