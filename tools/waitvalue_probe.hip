// Probe (round 6): does a plain device store into hipMallocSignalMemory release a hipStreamWaitValue64 on another stream, and
// how long after the store does the kernel behind the wait start?  It does, 2.4-3.4 us later -- but a kernel trace shows HOW:
// the runtime (ROCm 7.2) enqueues a polling kernel of its own, __amd_rocclr_streamOpsWait, i.e. a wave on the chip like the
// library's k_gate, not a wait packet of the queue's packet processor.  Not used by the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_writer(uint64_t *sig, uint64_t v, uint64_t ticks, unsigned long long *t)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    t[0] = __builtin_amdgcn_s_memrealtime();
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(sig), "v"(v));
    // stay on the chip a little: the kernel behind the wait must start while this one runs
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t1 < 5000) __builtin_amdgcn_s_sleep(8);
    t[2] = __builtin_amdgcn_s_memrealtime();
}
__global__ void k_after(unsigned long long *t) { t[1] = __builtin_amdgcn_s_memrealtime(); }
int main()
{
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint64_t *sig = nullptr;
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    *sig = 0;
    unsigned long long *t; CK(hipMalloc(&t, 24 * 64)); CK(hipMemset(t, 0, 24 * 64));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int i = 0; i < 32; ++i) {
        CK(hipStreamWaitValue64(b, sig, (uint64_t)(i + 1), hipStreamWaitValueGte));
        hipLaunchKernelGGL(k_after, dim3(1), dim3(1), 0, b, t + 3 * i);
        hipLaunchKernelGGL(k_writer, dim3(1), dim3(1), 0, a, sig, (uint64_t)(i + 1), 10000ull, t + 3 * i);
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    }
    unsigned long long h[3 * 32]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i)
        printf("%2d: store -> kernel behind the wait starts %+7.2f us (writer ended %+7.2f us after its store)\n", i,
               ((double)h[3 * i + 1] - (double)h[3 * i]) * 0.01, ((double)h[3 * i + 2] - (double)h[3 * i]) * 0.01);
    return 0;
}
