#!/bin/bash
# A diagnostic variant of the library next to the product: tools/build_variant.sh NAME "-DFLAG=1 ..." copies csrc/ to
# build/csrc_NAME, builds it with the extra flags (Makefile: EXTRA) and leaves build/libamrdemod_NAME.so, which travels to the
# GPU box with the snapshot; AMR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/libamrdemod_NAME.so makes the Python side load it.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
rm -rf $R/build/csrc_$N; mkdir -p $R/build/csrc_$N
cp $R/rtlamr_amd/csrc/*.h $R/rtlamr_amd/csrc/*.hip $R/rtlamr_amd/csrc/*.inc $R/rtlamr_amd/csrc/Makefile $R/build/csrc_$N/
make -C $R/build/csrc_$N -j8 EXTRA="$*" 2>&1 | grep -E "error|warning: unused|Error" || true
cp $R/build/csrc_$N/libamrdemod.so $R/build/libamrdemod_$N.so
rm -rf $R/build/csrc_$N
ls -la $R/build/libamrdemod_$N.so
