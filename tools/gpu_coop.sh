#!/bin/bash
# the cooperative K1 for partial wave-tiles: suite, single-block rate, a soak
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -n 4 2>&1 | tail -12
python tools/single_block_rate.py 2>&1 | tail -2
tools/gpu_soak.sh ${1:-1500} 8 > /dev/null 2>&1; tail -3 gpurun_out/soak_r03.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
