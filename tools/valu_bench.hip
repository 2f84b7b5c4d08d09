// Developer microbenchmark: VALU / LDS issue rates on gfx950 at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float *out, int iters, uint64_t *cyc)
{
    __shared__ float lut[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lut[i] = i;
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.5f;
    float2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, pb = {1.5f, 2.5f};
    uint32_t u0 = threadIdx.x, u1 = 77;
    uint32_t addr = (threadIdx.x & 31) * 4;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 32 independent v_add_f32 (8 chains)
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (MODE == 1) {  // 32 dependent v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
        } else if (MODE == 2) {  // 32 independent v_pk_add_f32 (4 chains)
            asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (MODE == 3) {  // 2-chain dependent v_add (distance 2)
            asm volatile(REP8("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(b));
        } else if (MODE == 4) {  // K1-like dependent group: add, sub, sub, alignbit
            asm volatile(REP8("v_add_f32 %0, %0, %3\n v_sub_f32 %1, %0, %3\n v_sub_f32 %1, %3, %1\n v_alignbit_b32 %2, %2, %1, 31\n")
                         : "+v"(a0), "+v"(a1), "+v"(u0) : "v"(b));
        } else if (MODE == 5) {  // 16 ds_read_b32 + wait
            float r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:128\n ds_read_b32 %2, %8 offset:256\n ds_read_b32 %3, %8 offset:384\n"
                         "ds_read_b32 %4, %8 offset:512\n ds_read_b32 %5, %8 offset:640\n ds_read_b32 %6, %8 offset:768\n ds_read_b32 %7, %8 offset:896\n"
                         "ds_read_b32 %0, %8 offset:1024\n ds_read_b32 %1, %8 offset:1152\n ds_read_b32 %2, %8 offset:1280\n ds_read_b32 %3, %8 offset:1408\n"
                         "ds_read_b32 %4, %8 offset:1536\n ds_read_b32 %5, %8 offset:1664\n ds_read_b32 %6, %8 offset:1792\n ds_read_b32 %7, %8 offset:1920\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
            a0 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        } else if (MODE == 6) {  // 32 sdwa shifts
            asm volatile(REP8("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
                              "v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                              "v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                              "v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n")
                         : "+v"(u0) : "v"(u1), "v"(2));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + u0;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char *name, int ninstr)
{
    float *o; uint64_t *c; hipMalloc(&o, 256 * 1024 * 4 * 4); hipMalloc(&c, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: block = 256*wps threads, one block per CU
        float best = 1e9; uint64_t cy = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, o, iters, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        double ns = best * 1e6 / ((double)wps * iters * ninstr);
        printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz), timer ticks/iter %.1f\n", name, wps, best,
               ns, ns * 2.4, (double)cy / iters);
    }
}
int main()
{
    run<0>("v_add_f32 indep x32", 32);
    run<1>("v_add_f32 dependent x32", 32);
    run<2>("v_pk_add_f32 indep x32", 32);
    run<3>("v_add_f32 2-chain x32", 32);
    run<4>("add,sub,sub,alignbit dep x32", 32);
    run<5>("ds_read_b32 x16 + wait", 16);
    run<6>("sdwa lshl x32", 32);
    return 0;
}
