#!/bin/bash
# Soak of the random-configuration parity tests on the GPU box: tools/gpu_soak.sh <seeds> [xdist workers]
# every seed = 4 tests (configuration, pipeline + deferral + validation, r900 digits, sharding), each against the oracle
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/soak_${TAG:-r04}.txt
N=${1:-3000}; W=${2:-6}
{ echo "# AMR_RANDOM_SEEDS=$N python -m pytest tests/test_gpu_random.py -n $W   ($(date -u +%FT%TZ), $(rocminfo 2>/dev/null | grep -m1 'Marketing Name.*MI' | sed 's/.*: *//'))"
  echo "# library: $(sha256sum rtlamr_amd/csrc/libamrdemod.so | cut -c1-16)  oracle: $(sha256sum oracle/decode_oracle.c | cut -c1-16)"; } > $O
t0=$(date +%s)
AMR_RANDOM_SEEDS=$N timeout 2400 python -m pytest tests/test_gpu_random.py -q -p no:cacheprovider -n $W 2>&1 | tail -15 >> $O
echo "# exit $? after $(( $(date +%s) - t0 )) s" >> $O
cat $O
