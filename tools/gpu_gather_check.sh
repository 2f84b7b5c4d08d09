#!/bin/bash
# both gather protocols on one box: the comm tests, then the bench as a 1-rank RCCL job with validated (whole slot) and raw (two-phase) records
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_deferral.py -q -m gpu > $O/pytest_comm.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest_comm.log
AMR_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; echo "validated rc=$?"
AMR_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 10 --gather raw > $O/bench_rccl_1rank_raw.json 2> $O/bench_rccl_1rank_raw.err; echo "raw rc=$?"
timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench_plain_same_box.json 2>/dev/null; echo "plain rc=$?"
python - <<'PY'
import json
for f in ("bench_rccl_1rank", "bench_rccl_1rank_raw", "bench_plain_same_box"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("gather_bytes_per_step"), d["config"].get("gather_records"))
    except Exception as e:
        print(f, "ERR", e)
PY
for f in $O/*.err; do tail -n 2 $f; done
