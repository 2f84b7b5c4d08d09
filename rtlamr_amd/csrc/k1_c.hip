#define AMR_K1_UNIT launch_k1_c
#define AMR_K1_CASES(X) X(80) X(88) X(96)
#ifdef AMR_K1T_DUMP_C
#define AMR_K1T_DUMP_HERE 1     // diagnostic builds: the timeline dump of this unit's kernels (chip 80 .. 96) instead of k1_b.hip's
#endif
#include "k1_launch.inc"
namespace amr {
bool launch_k1_a(int, dim3, hipStream_t, const K1Args &, hipEvent_t, hipEvent_t);
bool launch_k1_b(int, dim3, hipStream_t, const K1Args &, hipEvent_t, hipEvent_t);
bool launch_k1(int cl, dim3 grid, hipStream_t st, const K1Args &a, hipEvent_t start, hipEvent_t stop)
{
    return launch_k1_a(cl, grid, st, a, start, stop) || launch_k1_b(cl, grid, st, a, start, stop) || launch_k1_c(cl, grid, st, a, start, stop);
}
}  // namespace amr
