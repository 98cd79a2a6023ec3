// Device-side frame preprocessing (SURVEY.md par. 8f.1): the transform the reference applies on the
// host to every camera/video frame (src/utils/dataloader.py:18-32, src/real_time_inference.py:16-28):
//   ToTensor (uint8 HWC -> fp32 CHW / 255) -> Resize(224, bicubic; tensor path of torchvision 0.16 =
//   F.interpolate(mode='bicubic', align_corners=False, antialias=False), shorter side -> 224) ->
//   CenterCrop(224) -> BGR->RGB -> Normalize(CLIP mean/std)
// fused into one HBM-bound kernel that writes either the NCHW fp32 layout the reference's callers hand over
// (gitcap_preprocess) or, fused with the patch gather (SURVEY.md par. 8f.1), the bf16 patch rows the patch-embedding
// GEMM reads (gitcap_encode_raw / gitcap_greedy_raw: no fp32 frame tensor is materialised).
// One thread per output pixel (x fastest -> coalesced 4-byte stores per channel plane; the 4x4 taps
// of neighbouring threads overlap in L1/L2).  Bicubic follows ATen's upsample_bicubic2d exactly:
// A = -0.75, source index scale*(dst+0.5)-0.5, taps clamped to the image, x pass then y pass.
#include "kernels.h"

namespace {

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// PATCHES: out = bf16 patch rows [nf*G*G][Kp], k = c*ps*ps + py*ps + px, ps = patch size (the im2col layout; pad columns k >= 3 ps^2 are
// never written and stay zero from the allocation); else fp32 NCHW.
template <bool PATCHES>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ in, void* __restrict__ outp,
                                                         int nf, int H, int W, int crop, int newH, int newW,
                                                         int top, int left, float sy, float sx, int ps, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)nf * crop * crop;
    if (idx >= total) return;
    const int ox = (int)(idx % crop), oy = (int)((idx / crop) % crop);
    const int64_t f = idx / ((int64_t)crop * crop);
    const float A = -0.75f;
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float istd[3] = {1.f / 0.26862954f, 1.f / 0.26130258f, 1.f / 0.27577711f};
    float v[3];
    const unsigned char* img = in + f * (int64_t)H * W * 3;
    if (newH == H && newW == W) {                         // Resize is the identity: no interpolation at all
        const unsigned char* p = img + ((int64_t)(oy + top) * W + (ox + left)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)p[c] / 255.f;
    } else {
        const float ry = sy * ((float)(oy + top) + 0.5f) - 0.5f, rx = sx * ((float)(ox + left) + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        const int iy = (int)fy, ix = (int)fx;
        const float ty = ry - fy, tx = rx - fx;
        const float wx[4] = {cc2(tx + 1.f, A), cc1(tx, A), cc1(1.f - tx, A), cc2(2.f - tx, A)};
        const float wy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(2.f - ty, A)};
        float rowv[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), H - 1);
            float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), W - 1);
                const unsigned char* p = img + ((int64_t)yy * W + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) a[c] += ((float)p[c] / 255.f) * wx[j];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) rowv[i][c] = a[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = rowv[0][c] * wy[0] + rowv[1][c] * wy[1] + rowv[2][c] * wy[2] + rowv[3][c] * wy[3];
    }
    if (PATCHES) {
        const int G = crop / ps;
        bf16_t* o = (bf16_t*)outp + ((f * G + oy / ps) * G + ox / ps) * (int64_t)Kp + (oy % ps) * ps + (ox % ps);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * ps * ps] = f2bf((v[2 - c] - mean[c]) * istd[c]);   // same fp32 value, same rounding as im2col
    } else {
        float* o = (float*)outp + f * 3 * (int64_t)crop * crop + (int64_t)oy * crop + ox;
#pragma unroll
        for (int c = 0; c < 3; ++c)                            // output channel c (RGB) = input channel 2-c (BGR)
            o[(int64_t)c * crop * crop] = (v[2 - c] - mean[c]) * istd[c];
    }
}

}  // namespace

static hipError_t launch_pre(const unsigned char* in, void* out, int nf, int H, int W, int crop, int p, int Kp, hipStream_t s) {
    if (nf <= 0 || H <= 0 || W <= 0 || crop <= 0) return hipErrorInvalidValue;
    // torchvision _compute_resized_output_size: the shorter side becomes `crop`
    int newH, newW;
    if (H <= W) { newH = crop; newW = (int)((int64_t)crop * W / H); }
    else { newW = crop; newH = (int)((int64_t)crop * H / W); }
    if (newH < crop || newW < crop) return hipErrorInvalidValue;
    // CenterCrop: int(round((h - crop) / 2.0)), Python round = half to even
    auto half_even = [](int d) { const int q = d / 2; return (d & 1) ? ((q & 1) ? q + 1 : q) : q; };
    const int top = half_even(newH - crop), left = half_even(newW - crop);
    const float sy = (float)H / (float)newH, sx = (float)W / (float)newW;
    const int64_t total = (int64_t)nf * crop * crop;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (p > 0) hipLaunchKernelGGL(preprocess_kernel<true>, grid, dim3(256), 0, s, in, out, nf, H, W, crop, newH, newW, top, left, sy, sx, p, Kp);
    else hipLaunchKernelGGL(preprocess_kernel<false>, grid, dim3(256), 0, s, in, out, nf, H, W, crop, newH, newW, top, left, sy, sx, 1, 0);
    return hipGetLastError();
}

hipError_t launch_preprocess(const unsigned char* in, float* out, int nf, int H, int W, int crop, hipStream_t s) {
    return launch_pre(in, out, nf, H, W, crop, 0, 0, s);
}

hipError_t launch_preprocess_patches(const unsigned char* in, bf16_t* patches, int nf, int H, int W, int crop, int p, int Kp, hipStream_t s) {
    if (p <= 0 || crop % p || Kp < 3 * p * p) return hipErrorInvalidValue;
    return launch_pre(in, patches, nf, H, W, crop, p, Kp, s);
}
