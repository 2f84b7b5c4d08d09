#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh pmc_cfg3 cfg3 1 > /dev/null 2>&1
bash tools/gpu_pmc.sh pmc_cfg5 cfg5 1 > /dev/null 2>&1
bash tools/gpu_pmc.sh pmc_cfg2 cfg2 1 > /dev/null 2>&1
