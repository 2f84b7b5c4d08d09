// Developer microbenchmark: latency of an exact sequential float32 prefix sum over the 64 lanes of a wave (the r900
// second-stage running sum, k4_r900.h), by implementation.  Prints shader cycles per 64-sample tile, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, const float *in, int tiles, unsigned long long *cyc)
{
    const int lane = threadIdx.x;
    float m = in[lane] + 1.0f;
    float carry = 0.f, acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        float p;
        if (MODE == 0) {          // DPP wave shift, 2 wait states
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(p) : "v"(carry), "v"(m));
#pragma unroll
            for (int i = 0; i < 63; ++i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(m));
        } else if (MODE == 1) {   // DPP row shift (wrong sums across rows: timing only)
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(p) : "v"(carry), "v"(m));
#pragma unroll
            for (int i = 0; i < 63; ++i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(m));
        } else if (MODE == 2) {   // every lane adds the broadcast magnitudes in order, lane i stops after sample i (exec shrinks)
            p = carry;
            const unsigned long long all = __builtin_amdgcn_read_exec();
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const float mi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), i));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(p) : "s"(mi));
                if (i < 63) asm volatile("s_lshl_b64 exec, exec, 1" ::: "memory");
            }
            asm volatile("s_mov_b64 exec, %0" :: "s"(all) : "memory");
        } else if (MODE == 3) {   // all lanes compute every prefix, lane i keeps prefix i (cndmask)
            float s = carry; p = 0.f;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const float mi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), i));
                s += mi;
                p = lane == i ? s : p;
            }
        } else if (MODE == 5) {   // round 4: every lane runs the SAME chain on scalar operands (64 readlanes up front), the running
                                  // sum goes to LDS step by step (all lanes store the same dword) -- the filter phase reads sums from LDS anyway
            __shared__ float ring5[64];
            int ms[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) ms[i] = __builtin_amdgcn_readlane(__float_as_int(m), i);
            p = carry;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(p) : "s"(ms[i]));
                asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(0), "v"(p), "n"(i * 4) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            p = ring5[lane];      // lane i: prefix i (what the DPP chain leaves in the register)
        } else if (MODE == 6) {   // as 5, readlane and add interleaved (the readlane of step i+1 issues behind the add of step i)
            __shared__ float ring6[64];
            p = carry;
            int mi = __builtin_amdgcn_readlane(__float_as_int(m), 0);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int mn = __builtin_amdgcn_readlane(__float_as_int(m), i < 63 ? i + 1 : 63);
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(p) : "s"(mi));
                asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(0), "v"(p), "n"(i * 4) : "memory");
                mi = mn;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            p = ring6[lane];
        } else {                  // plain dependent adds (lower bound)
            p = carry;
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p) : "v"(m));
        }
        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));
        acc += p;
        m += 0.25f;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc;
    if (blockIdx.x == 0 && lane == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name)
{
    float *o, *in; unsigned long long *c;
    hipMalloc(&o, 1024 * 64 * 4); hipMalloc(&in, 256); hipMalloc(&c, 8); hipMemset(in, 0, 256);
    const int tiles = 400;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE>), dim3(1024), dim3(64), 0, 0, o, in, tiles, c);
    hipDeviceSynchronize();
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    float h[64]; hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
    printf("%-40s %7.1f cycles per 64-sample tile  (lane 5 acc %.3f, lane 40 acc %.3f)\n", name, (double)cy / tiles, h[5], h[40]);
}
int main()
{
    run<4>("plain dependent v_add x64 (floor)");
    run<0>("v_add_dpp wave_shr:1 + s_nop 1");
    run<1>("v_add_dpp row_shr:1 + s_nop 1 (timing)");
    run<2>("readlane + v_add under shrinking exec");
    run<3>("readlane + v_add + cndmask");
    run<5>("64 readlanes, then uniform v_add(sgpr) + ds_write per step");
    run<6>("readlane / v_add(sgpr) / ds_write interleaved");
    return 0;
}
