// Co-residency probe (tools/coresidency.py): single-wave streaming kernels that differ only in their VGPR
// footprint (NV float4 per lane held live), launched next to the 256x256 GEMM on another stream.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NV>
__global__ __launch_bounds__(64) void probe_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* p = in + ((size_t)blockIdx.x * iters * NV * 64 + lane) * 4;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(p + (size_t)(it * NV + i) * 256);
#pragma unroll
        for (int i = 0; i < NV; ++i) acc += v[i];
    }
    *(f32x4*)(out + ((size_t)blockIdx.x * 64 + lane) * 4) = acc;
}
// CU lock-out without memory traffic: 1024-thread workgroups (16 waves, padded to > 96 VGPRs like attn_text) that sleep
__global__ __launch_bounds__(1024) void lock_kernel(float* out, int sleeps) {
    float keep[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) keep[i] = out[(threadIdx.x + i * 1024) & 4095];
    asm volatile("v_mov_b32 v111, 0" ::: "v111");          // allocate 112 VGPRs per wave, like attn_text
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += keep[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
extern "C" int lock_launch(float* out, int blocks, int sleeps, void* stream) {
    hipLaunchKernelGGL(lock_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, out, sleeps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int probe_launch(int nv, const float* in, float* out, int blocks, int floats_per_block, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int iters = floats_per_block / (nv * 256);
    if (nv == 4) hipLaunchKernelGGL(probe_kernel<4>, dim3(blocks), dim3(64), 0, s, in, out, iters);
    else if (nv == 8) hipLaunchKernelGGL(probe_kernel<8>, dim3(blocks), dim3(64), 0, s, in, out, iters);
    else if (nv == 16) hipLaunchKernelGGL(probe_kernel<16>, dim3(blocks), dim3(64), 0, s, in, out, iters);
    else if (nv == 24) hipLaunchKernelGGL(probe_kernel<24>, dim3(blocks), dim3(64), 0, s, in, out, iters);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
