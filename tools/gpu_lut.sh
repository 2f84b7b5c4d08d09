#!/bin/bash
# A/B of the LUT fill (AMR_K1T_LUT_DMA 0 / 1 harness builds) + parity and bench with the product library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/lut.log; : > $O
tools/gpu_k1h.sh > gpurun_out/k1h_out.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py tests/test_gpu_deferral.py -x -q > gpurun_out/lut_tests.log 2>&1; echo "tests exit $?" >> $O
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "cfg2-shard0 or cfg4:8 or cfg4:40 or cfg4:88" >> gpurun_out/lut_tests.log 2>&1; echo "fullsize exit $?" >> $O
for w in cfg2 cfg4:8 cfg4:40 cfg2; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> $O; done
tail -3 gpurun_out/lut_tests.log; cat gpurun_out/k1h_out.log
python - <<'PY'
import json
for l in open('gpurun_out/lut.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        bad=[k for k,v in j['config']['checks'].items() if 'MISMATCH' in str(v)]
        print(j['config']['workload'][:30], f"{j['value']/1e6:.3f}e6 step {j['ms_per_step']:.4f} steady {j['steady_ms_per_step']:.4f} k1 {r['k1_ms']:.4f} frac {r['frac']:.3f}", 'BAD' if bad else 'ok')
    else: print(l.strip())
PY
