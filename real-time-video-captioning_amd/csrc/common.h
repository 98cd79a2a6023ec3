// Shared device/host helpers for libgitcap (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// fp32 -> bf16 round-to-nearest-even.  A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaN a NaN (MI355X_MICROARCH.md, correctness boundaries).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
    return __builtin_bit_cast(float, (unsigned)h << 16);
}
// 8 OCP e4m3 bytes x one power-of-two scale -> one bf16 MFMA fragment: four v_cvt_scalef32_pk_bf16_fp8 (gfx950).
// Exact: e4m3 x 2^k is a bf16 value, so the fragment is bit for bit the bf16-stored weight.
__device__ __forceinline__ bf16x8 fp8x8_to_bf16x8(uint2 q, float scale) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, scale, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, scale, true);
    const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, scale, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, scale, true);
    const u32x4_t r = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, v);
}

// four floats -> four OCP e4m3 codes (round to nearest even, saturating at +-448): two v_cvt_pk_fp8_f32
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}

// e4m3 codes that pack_fp8x4 clamps (|value| beyond the largest finite e4m3): the saturation the fp8 activation path must
// not hide (gitcap_fp8_saturations)
__device__ __forceinline__ int count_fp8_clamped(f32x4 c) {
    return (int)(fabsf(c[0]) > 448.f) + (int)(fabsf(c[1]) > 448.f) + (int)(fabsf(c[2]) > 448.f) + (int)(fabsf(c[3]) > 448.f);
}

// 16-byte-chunk XOR swizzle for [rows][64 bf16] (128-B row) LDS tiles read with ds_read_b128 by
// MFMA operand lanes (row = lane&15 or lane&31, chunk = k/8).  g(row) = (row>>1)&7 makes every
// ds_read_b128 lane group hit 16 distinct 16-B slots of the 256-B bank row for both the
// 16x16x32 and the 32x32x16 operand maps (derivation in docs/LAB_NOTEBOOK.md, "LDS images").
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// XCD-aware bijective block remap (cdna_hip_programming.md T1): blocks that share an XCD
// (bid % 8 equal) get a contiguous run of logical tile ids, so tiles that share an operand
// panel hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// one (non-returning) atomic per wave that clamped at least one e4m3 code: total += the wave's count
__device__ __forceinline__ void report_fp8_clamped(unsigned long long* total, int nsat, int lane) {
    if (!total) return;
    const unsigned long long any = __ballot(nsat != 0);
    if (any == 0) return;
    const int n = (int)wave_sum((float)nsat);              // <= 64 lanes x 128 codes: exact in fp32
    if (lane == 0) (void)__hip_atomic_fetch_add(total, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The library is compiled with -ffp-contract=off: the compiler never fuses a multiply with an add on its own, so the
// rounding of every fp32 expression is what the source says, in every kernel and every instantiation (three tile kernels
// share these epilogue functions and must agree bit for bit; batch invariance rests on it).  Where a fused multiply-add is
// wanted it is written as __builtin_fmaf.
// x * sigmoid(1.702 x) with v_exp_f32 / v_rcp_f32 (each ~1 ulp; the result is rounded to bf16)
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the output):
// erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1/(1 + p z), z >= 0; odd extension.
__device__ __forceinline__ float fast_erf(float x) {
    const float z = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float r = __builtin_fmaf(-(p * t), e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float erf_gelu(float x) {
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, fast_erf(x * 0.70710678118654752f), hx);
}
