#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
(timeout 600 python bench.py --steps 5 --warmup 2) > gpurun_out/bench1.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench1.log
(timeout 300 python -c "
import torch, sys
sys.path.insert(0,'.')
print(torch.cuda.is_available(), torch.cuda.get_device_name(0))
import __graft_entry__ as g; g.smoke()
") > gpurun_out/torch_coexist.log 2>&1
echo "rc=$?" >> gpurun_out/torch_coexist.log
tail -5 gpurun_out/smoke.log gpurun_out/bench1.log gpurun_out/torch_coexist.log
