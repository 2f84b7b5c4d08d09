// the fallback search kernels of k2_search.h: preambles shorter than 16 symbols, more than four preambles, rows under 16
// words, and the re-run after a candidate-list overflow
#include "launch.h"
#include "k2_search.h"

namespace amr {

bool launch_k2_fast(uint32_t n_pre, int nwv, uint32_t grid, size_t lds, hipStream_t st, hipEvent_t start, hipEvent_t stop,
                    const K2Args &a, hipError_t *err)
{
#define AMR_K2_LAUNCH(N, W, J)                                                                                           \
    do {                                                                                                               \
        *err = hipFuncSetAttribute((const void *)k2_search_fast<N, W, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (*err == hipSuccess)                                                                                        \
            hipExtLaunchKernelGGL((k2_search_fast<N, W, J>), dim3(grid), dim3(64 * W), lds, st, start, stop, 0, a);      \
    } while (0)
#define AMR_K2_CASE(N)                                                                                                   \
    case N:                                                                                                            \
        if (nwv == 8) AMR_K2_LAUNCH(N, 8, 8);   /* 16 words per step measured slower (51 vs 47 us) */                  \
        else AMR_K2_LAUNCH(N, 4, 4);                                                                                   \
        return true;
    switch (n_pre) { AMR_K2_CASE(1) AMR_K2_CASE(2) AMR_K2_CASE(3) AMR_K2_CASE(4) }
#undef AMR_K2_CASE
#undef AMR_K2_LAUNCH
    return false;
}

void launch_k2_dense(uint32_t grid, size_t lds, hipStream_t st, hipEvent_t start, hipEvent_t stop, const K2Args &a, hipError_t *err)
{
    *err = hipFuncSetAttribute((const void *)k2_search_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (*err == hipSuccess) hipExtLaunchKernelGGL(k2_search_dense, dim3(grid), dim3(256), lds, st, start, stop, 0, a);
}

}  // namespace amr
