#!/bin/bash
# Phases of K3's workgroups (100 MHz stamps of a -DAMR_K3_DBG=1 build of the library, printed by amr_destroy), plain and
# validated, bench --depth 1 (nothing runs next to K3).  Build the diagnostic library first:
#   mkdir -p build/csrc_k3dbg && cp rtlamr_amd/csrc/{*.h,*.hip,*.inc,Makefile} build/csrc_k3dbg/ &&
#   sed -i 's|-I../../include|-I/root/repo/include -DAMR_K3_DBG=1|; s|\.\./\.\./include/amrdemod.h|/root/repo/include/amrdemod.h|' build/csrc_k3dbg/Makefile &&
#   make -C build/csrc_k3dbg -j8
cd $GRAFT_REPO_ROOT; L=$GRAFT_REPO_ROOT/build/csrc_k3dbg/libamrdemod.so
for w in ${WORKLOADS:-cfg2}; do for m in "" --validate; do
echo "== $w ${m:-plain}"
AMR_LIB_OVERRIDE=$L timeout 200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-verify --depth 1 --k1-events 0 $m 2>&1 >/dev/null | grep -A8 K3_DBG | head -9
done; done
