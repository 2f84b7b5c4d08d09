#!/bin/bash
# validated mode (K5 as K3's last stage + k5_compact): its tests, the random sweep, the bench and the timeline of the tail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_validate.py tests/test_gpu_comm.py tests/test_gpu_deferral.py -q -m gpu -x > $O/pytest_val.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_val.log
AMR_RANDOM_SEEDS=${SEEDS:-400} timeout 900 python -m pytest tests/test_gpu_random.py -q -m gpu -x -n 8 > $O/pytest_val_random.log 2>&1; echo "random rc=$?"; tail -n 2 $O/pytest_val_random.log
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --validate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('validate', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   ', d['value'], d['ms_per_step'])"
done
for w in cfg3 cfg5; do
timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --validate 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w validate', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w plain   ', d['value'], d['ms_per_step'])"
done
tools/gpu_timeline.sh val --validate | head -42 | tail -18
