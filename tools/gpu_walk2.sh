#!/bin/bash
# call-aligned walk: harness + parity tests + bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/walk2.log; : > $O
tools/gpu_k2b.sh k2b > gpurun_out/k2b_out.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 > gpurun_out/walk2_tests.log 2>&1; echo "tests exit $?" >> $O
for w in cfg2 cfg3 cfg5 cfg4:8 cfg2; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> $O; done
tail -5 gpurun_out/walk2_tests.log; grep -v 'cycles' gpurun_out/k2b_out.log
python - <<'PY'
import json
for l in open('gpurun_out/walk2.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        bad=[k for k,v in j['config']['checks'].items() if 'MISMATCH' in str(v)]
        print(j['config']['workload'][:30], f"{j['value']/1e6:.3f}e6 step {j['ms_per_step']:.4f} steady {j['steady_ms_per_step']:.4f} k1 {r['k1_ms']:.4f} frac {r['frac']:.3f} search {r['search_ms']:.4f} whole {r['whole_path_frac']:.3f}", 'BAD' if bad else 'ok')
    else: print(l.strip())
PY
