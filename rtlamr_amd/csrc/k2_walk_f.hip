#define AMR_K2W_UNIT launch_k2_walk_f
#define AMR_K2W_SLS(X) X(192)
#include "k2_walk_launch.inc"
namespace amr {
bool launch_k2_walk_a(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_walk_b(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_walk_c(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_walk_d(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_walk_e(uint32_t, uint32_t, uint32_t, size_t, hipStream_t, hipEvent_t, hipEvent_t, const K2Args &, hipError_t *);
bool launch_k2_walk(uint32_t sl, uint32_t set, uint32_t grid, size_t lds, hipStream_t st, hipEvent_t start, hipEvent_t stop, const K2Args &a,
                    hipError_t *err)
{
    return launch_k2_walk_a(sl, set, grid, lds, st, start, stop, a, err) ||
           launch_k2_walk_b(sl, set, grid, lds, st, start, stop, a, err) ||
           launch_k2_walk_c(sl, set, grid, lds, st, start, stop, a, err) ||
           launch_k2_walk_d(sl, set, grid, lds, st, start, stop, a, err) ||
           launch_k2_walk_e(sl, set, grid, lds, st, start, stop, a, err) ||
           launch_k2_walk_f(sl, set, grid, lds, st, start, stop, a, err);
}
int k2_walk_kind_of(uint32_t len, uint64_t bits) { return k2_walk_kind(len, bits); }
}  // namespace amr
