#!/bin/bash
# usage: tools/variant_check.sh <diag id>: parity tests (core GPU files) + A/B bench of build/diag<N>.so against the product build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export AMR_LIB_OVERRIDE=$PWD/build/diag$1.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
  AMR_LIB_OVERRIDE=$PWD/build/diag$1.so python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('variant', j['value'], j['ms_per_step'], j['roofline']['k1_ms'])"
  env -u AMR_LIB_OVERRIDE python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('product', j['value'], j['ms_per_step'], j['roofline']['k1_ms'])"
done
