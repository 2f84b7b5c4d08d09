#define AMR_K2R_UNIT launch_k2_row_d
#define AMR_K2R_SLS(X) X(160) X(176) X(192)
#include "k2_row_launch.inc"
namespace amr {
bool launch_k2_row_a(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_row_b(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_row_c(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_row(uint32_t sl, uint32_t kind, uint32_t grid, size_t lds, hipStream_t st, hipEvent_t start, hipEvent_t stop, const K2Args &a,
                   hipError_t *err)
{
    return launch_k2_row_a(sl, kind, grid, lds, st, start, stop, a, err) || launch_k2_row_b(sl, kind, grid, lds, st, start, stop, a, err) ||
           launch_k2_row_c(sl, kind, grid, lds, st, start, stop, a, err) || launch_k2_row_d(sl, kind, grid, lds, st, start, stop, a, err);
}
}  // namespace amr
