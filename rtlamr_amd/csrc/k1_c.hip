#define AMR_K1_UNIT launch_k1_c
#define AMR_K1_CASES(X) X(80) X(88) X(96)
#include "k1_launch.inc"
namespace amr {
bool launch_k1_a(int, bool, dim3, hipStream_t, const K1Args &, hipEvent_t, hipEvent_t);
bool launch_k1_b(int, bool, dim3, hipStream_t, const K1Args &, hipEvent_t, hipEvent_t);
bool launch_k1(int cl, bool tail, dim3 grid, hipStream_t st, const K1Args &a, hipEvent_t start, hipEvent_t stop)
{
    return launch_k1_a(cl, tail, grid, st, a, start, stop) || launch_k1_b(cl, tail, grid, st, a, start, stop) ||
           launch_k1_c(cl, tail, grid, st, a, start, stop);
}
}  // namespace amr
