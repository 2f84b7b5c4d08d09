// amr_util.hip -- device utilities and the synthetic IQ generator for bench and tests; pinned host
// buffers for amr_submit_host.  Not part of the decode path.
#include "amr_host.h"
#include "synth.h"

using namespace amr_host;

extern "C" {

amr_status amr_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr || bytes == 0) return fail(AMR_EINVAL, "null argument");
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(AMR_ENOMEM, "hipHostMalloc", e);
    return AMR_OK;
}

amr_status amr_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return AMR_OK;
}

amr_status amr_device_count(int32_t *n_devices)
{
    if (!n_devices) return fail(AMR_EINVAL, "null argument");
    *n_devices = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AMR_OK;    // none: not an error, the count is the answer
    int n = 0;
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++n;
    }
    *n_devices = n;
    return AMR_OK;
}
/* ---- device utilities ---- */

amr_status amr_dev_alloc(int32_t device_id, size_t bytes, void **d_ptr)
{
    if (!d_ptr) return fail(AMR_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(device_id));
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) return fail(AMR_ENOMEM, "hipMalloc", e);
    return AMR_OK;
}
amr_status amr_dev_free(int32_t device_id, void *d_ptr)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipFree(d_ptr));
    return AMR_OK;
}
amr_status amr_dev_upload(int32_t device_id, void *d_dst, const void *src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return AMR_OK;
}
amr_status amr_dev_download(int32_t device_id, void *dst, const void *d_src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return AMR_OK;
}
amr_status amr_dev_sync(int32_t device_id)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipDeviceSynchronize());
    return AMR_OK;
}

static amr_status synth_fill(bool uniform, int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    if (!d_iq || (n_samples & 7)) return fail(AMR_EINVAL, "n_samples must be a multiple of 8");
    HIP_TRY(hipSetDevice(device_id));
    const uint64_t threads = n_samples / 8;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (uniform) hipLaunchKernelGGL(amr::k_synth_noise<true>, grid, dim3(256), 0, 0, (uint8_t *)d_iq, n_samples, seed, first_sample);
    else hipLaunchKernelGGL(amr::k_synth_noise<false>, grid, dim3(256), 0, 0, (uint8_t *)d_iq, n_samples, seed, first_sample);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return AMR_OK;
}

amr_status amr_synth_noise(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    return synth_fill(false, device_id, d_iq, n_samples, seed, first_sample);
}

amr_status amr_synth_uniform(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    return synth_fill(true, device_id, d_iq, n_samples, seed, first_sample);
}

amr_status amr_synth_plant(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t first_sample, int32_t chip_length,
                           uint32_t n_packets, const uint64_t *start, const uint8_t *bits, uint32_t n_bits,
                           uint32_t stride, const int8_t *d_i, const int8_t *d_q)
{
    if (!d_iq || !start || !bits || !d_i || !d_q || chip_length <= 0) return fail(AMR_EINVAL, "null argument");
    if (n_packets == 0) return AMR_OK;
    HIP_TRY(hipSetDevice(device_id));
    uint64_t *ds = nullptr; uint8_t *db = nullptr; int8_t *di = nullptr, *dq = nullptr;
    HIP_TRY(hipMalloc((void **)&ds, n_packets * 8ull));
    HIP_TRY(hipMalloc((void **)&db, (size_t)n_packets * stride));
    HIP_TRY(hipMalloc((void **)&di, n_packets));
    HIP_TRY(hipMalloc((void **)&dq, n_packets));
    HIP_TRY(hipMemcpy(ds, start, n_packets * 8ull, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, bits, (size_t)n_packets * stride, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di, d_i, n_packets, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dq, d_q, n_packets, hipMemcpyHostToDevice));
    amr::PlantArgs a{(uint8_t *)d_iq, n_samples, first_sample, ds, db, di, dq, n_packets, n_bits, stride,
                     (uint32_t)chip_length};
    const uint32_t per = n_bits * 2u * (uint32_t)chip_length;
    hipLaunchKernelGGL(amr::k_synth_plant, dim3((per + 255) / 256, n_packets), dim3(256), 0, 0, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(ds); (void)hipFree(db); (void)hipFree(di); (void)hipFree(dq);
    return AMR_OK;
}

}  // extern "C"
