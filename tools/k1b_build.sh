#!/bin/bash
# usage: tools/k1b_build.sh <tag> [extra -D flags...]: build build/k1b_<tag> and print the resource usage of the K1 kernels
tag=$1; shift
cd /root/repo && mkdir -p build/k1b_tmp_$tag && cd build/k1b_tmp_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../rtlamr_amd/csrc -I../../include -DK1B_CL=${K1B_CL:-72} "$@" -save-temps=obj -o ../k1b_$tag ../../tools/k1_bench.hip 2>&1 | grep -v 'warning\|^$' | grep -B2 -A6 'error' | head -40
mv ../k1_bench-hip-amdgcn-amd-amdhsa-gfx950.s . 2>/dev/null; rm -f ../k1_bench-h* ; S=k1_bench-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E '^_ZN3amr[0-9]+k1t?_demodILi[0-9]+ELb0.*:|; (NumVgprs|ScratchSize): ' $S | awk '/^_ZN/{n=$1} /NumVgprs/{v=$3} /ScratchSize/{ if (n!="") print n, "vgprs", v, "scratch", $3; n=""}'
