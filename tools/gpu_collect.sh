#!/bin/bash
cd $GRAFT_REPO_ROOT; tools/collect_profiles_r03.sh r03 > gpurun_out/collect_r03.log 2>&1; tail -12 gpurun_out/collect_r03.log
