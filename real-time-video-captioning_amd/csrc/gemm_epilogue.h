// LDS-staged epilogue shared by the 256x256 (gemm256.hip, gemm256p.hip uses its own) and 256x128 (gemm2b.hip)
// kernels: both give a wave the same 128(n) x 64(m) accumulator block acc[x][i][y][j]
// (n = x*64 + i*16 + 4*(lane>>4) + r, m = y*32 + j*16 + (lane&15)).
//
// The accumulator layout (lane = one m, 4 consecutive n) would scatter 32-byte pieces over 16 rows
// per store instruction.  Instead the wave transposes its two 64(m) x 64(n) half-blocks through a
// private LDS region (the operand stages are dead by then) and writes/reads global memory as full row
// segments: 16 B per lane, 128 B (bf16) or 256 B (fp32) per row.
#pragma once
#include "kernels.h"

constexpr int EPI_REGION = 64 * (64 * 4 + 16);   // per-wave staging (fp32 worst case): 17408 B

// ep: this wave's EPI_REGION bytes of LDS; mw / nw: first m / n of the wave's block
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_wave(const GemmArgs& a, const f32x4 (&acc)[2][4][2][2], char* ep,
                                                   const int mw, const int nw, const int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16);
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int RS = 64 * ESZ + 16;                      // padded row stride (bytes)
    constexpr int LPR = 64 * ESZ / 16;                     // lanes per row on the row-wise side (8 or 16)
    constexpr int RPI = 64 / LPR;                          // rows per wave-instruction (8 or 4)
    const int rr = lane / LPR, rc = lane % LPR;            // row-wise role of this lane
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int nb = nw + x * 64;                     // first n of this half-block
        // 1) accumulator layout -> LDS [m_local][n_local]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nl = i * 16 + fq * 4;
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_PATCH_F32 && a.bias) bias4 = *(const f32x4*)(a.bias + nb + nl);
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = y * 32 + j * 16 + frow;
                    f32x4 v = acc[x][i][y][j] + bias4;
                    if (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                    } else if (EPI == EPI_BIAS_GELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = erf_gelu(v[r]);
                    }
                    if (OUT_BF16) {
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        *(uint2*)(ep + ml * RS + nl * 2) = o;
                    } else {
                        *(f32x4*)(ep + ml * RS + nl * 4) = v;
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 2) LDS rows -> global, 16 B per lane along n
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int ml = it * RPI + rr;
            const int m = mw + ml;
            const int n = nb + rc * (16 / ESZ);
            const uint4 raw = *(const uint4*)(ep + ml * RS + rc * 16);
            if (OUT_BF16) {
                *(uint4*)((bf16_t*)a.out + (size_t)m * a.ldo + n) = raw;
            } else {
                f32x4 v = __builtin_bit_cast(f32x4, raw);
                if (EPI == EPI_BIAS_RESID_F32) {
                    v += *(const f32x4*)(a.resid + (size_t)m * a.ldr + n);
                    *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
                } else if (EPI == EPI_BIAS_F32) {
                    *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
                } else {  // EPI_PATCH_F32: m = frame*P + patch -> row frame*N + 1 + patch, + pos[1+patch]
                    if (m < a.valid_rows) {
                        const int frame = m / a.patches_per_frame;
                        const int patch = m - frame * a.patches_per_frame;
                        v += *(const f32x4*)(a.pos + (size_t)(1 + patch) * a.N + n);
                        const size_t orow = (size_t)frame * a.tokens_per_frame + 1 + patch;
                        *(f32x4*)((float*)a.out + orow * a.ldo + n) = v;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      // reads done before the next half-block overwrites
    }
}
