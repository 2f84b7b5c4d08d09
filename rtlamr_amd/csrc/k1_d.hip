// K1 for partial wave-tiles: one wave per block (k1_coop.h), every legal chip length (flags.go:127-132)
#include "launch.h"
#include "k1_coop.h"

namespace amr {

bool launch_k1_coop(int cl, uint32_t first_block, uint32_t n_blocks, hipStream_t st, K1Args a, hipEvent_t start, hipEvent_t stop)
{
    a.wg_first = first_block;      // a BLOCK index for this kernel
    switch (cl) {
#define AMR_K1C_CASE(N) case N: hipExtLaunchKernelGGL((k1c_demod<N>), dim3(n_blocks), dim3(64), 0, st, start, stop, 0, a); return true;
        AMR_K1C_CASE(8) AMR_K1C_CASE(32) AMR_K1C_CASE(40) AMR_K1C_CASE(48) AMR_K1C_CASE(56) AMR_K1C_CASE(64) AMR_K1C_CASE(72)
        AMR_K1C_CASE(80) AMR_K1C_CASE(88) AMR_K1C_CASE(96)
#undef AMR_K1C_CASE
    default: return false;
    }
}

}  // namespace amr
