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
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// 16-byte-chunk XOR swizzle for [rows][64 bf16] (128-B row) LDS tiles read with ds_read_b128 by
// MFMA operand lanes (row = lane&15 or lane&31, chunk = k/8).  g(row) = (row>>1)&7 makes every
// ds_read_b128 lane group hit 16 distinct 16-B slots of the 256-B bank row for both the
// 16x16x32 and the 32x32x16 operand maps (derivation in DESIGN.md, "LDS images").
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

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float erf_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
