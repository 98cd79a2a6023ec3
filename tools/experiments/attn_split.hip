// EXPERIMENT (measured, not taken; tools/probe/coattn_probe.hip): single-wave split form of the text-row attention.
// Idea: waves of <= 48 VGPRs and 6 KiB LDS fit BESIDE a 256x256 GEMM workgroup, so the token loop of batch i would stop
// evicting the image-pass GEMMs of batch i+1 from whole CUs.  Measured on MI355X (QKV-shaped GEMM train on one stream,
// attention launches on another): one attention launch per GEMM costs the GEMM 8 us in the 16-wave workgroup form
// and 9.5 us in this form; a dense attention train slows the GEMM by 28 % (workgroup form) and 43 % (this form).
// What the GEMM loses is HBM bandwidth to the 58 MB of K/V each launch streams, not CUs; alone on the chip this form
// is not faster either (21.2 vs 19.1 us).  Include after csrc/txtblock.hip (uses its helpers).
struct TxtSplitArgs {
    const bf16_t* kv_img; const bf16_t* kv_txt;
    int rows, beams, t0, T, Tmax, S_img, H, D;
    float* part_o;                              // [M*H][nchunk_max][64] fp32 chunk partials
    float* part_ml;                             // [M*H][nchunk_max][2]  (running max, sum)
    unsigned* cnt;                              // [M*H] arrival tickets (zero between launches)
    bf16_t* ctx;                                // out [M][D]
    int nchunk_max;                             // set by the launcher
};

// ------------------------------------------------------------------------------------------------------------------
// Split form of the text-row attention: ONE WAVE per (text row, head, 64-key chunk), built to run BESIDE the image
// pass of another batch.  A 256x256 GEMM workgroup leaves 48 VGPRs per SIMD lane and 24 KiB of LDS on its CU:
// a wave of this kernel (<= 48 VGPRs, 6 KiB LDS) fits next to it, so the token loop of batch i no longer evicts the
// GEMMs of batch i+1 from whole CUs (the 16-wave workgroup above needs a CU to itself).  K/V travel by LDS-DMA
// through a 3-slot ring (8 keys = 1 KiB of K + 1 KiB of V per slot): no registers are held by data in flight.
//   per lane: sub = lane & 7 owns dims 8 sub .. +7, kk = lane >> 3 takes keys kk, kk + 8, ... of the chunk (online
//   softmax); the 8 key groups are merged by shuffles; the chunk's (m, l, o[64]) goes to HBM (write-through);
//   the LAST chunk of a (row, head) to arrive merges all chunks IN CHUNK ORDER (batch invariant) and writes the
//   context row (bf16).  Same ticket protocol as above (Guideline 16; no spin).
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int SPLIT_KEYS = 64;                     // keys per wave

__global__ __launch_bounds__(64) void attn_split_kernel(TxtSplitArgs a) {
    __shared__ __attribute__((aligned(16))) char ring[3 * 2048];
    const int lane = threadIdx.x;
    const int D = a.D, ld = 3 * D;
    const int nc = a.nchunk_max;
    const int unit = blockIdx.x / nc, c = blockIdx.x - unit * nc;       // unit = m * H + head
    const int m = unit / a.H, head = unit - m * a.H;
    const int r = m / a.T, j = m - r * a.T;
    const int tq = a.t0 + j;
    const int Lk = a.S_img + tq + 1;
    const int k0 = c * SPLIT_KEYS;
    if (k0 >= Lk) return;                                                // (uniform) chunk past this unit's last key
    const int nchunks = (Lk + SPLIT_KEYS - 1) / SPLIT_KEYS;
    const int sub = lane & 7, kk = lane >> 3;
    const bf16_t* img = a.kv_img + (size_t)(r / a.beams) * a.S_img * ld + D + head * 64 + sub * 8;
    const bf16_t* txt = a.kv_txt + (size_t)r * a.Tmax * ld + D + head * 64 + sub * 8;

    auto dma_step = [&](int s) {                                         // keys k0 + 8 s + kk -> ring slot s % 3
        int key = k0 + s * 8 + kk;
        key = key < Lk ? key : 0;
        const bf16_t* kp = key < a.S_img ? img + (size_t)key * ld : txt + (size_t)(key - a.S_img) * ld;
        char* slot = ring + (s % 3) * 2048;
        __builtin_amdgcn_global_load_lds(GLB_PTR(kp), LDS_PTR(slot), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(kp + D), LDS_PTR(slot + 1024), 16, 0, 0);
    };
    dma_step(0);
    dma_step(1);

    float qv[8];
    {
        const bf16x8 q8 = *(const bf16x8*)(a.kv_txt + ((size_t)r * a.Tmax + tq) * ld + head * 64 + sub * 8);
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = bf2f((bf16_t)q8[d]) * kScaleLog2e;
    }
    float sm = -INFINITY, sl = 0.f, so[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) so[d] = 0.f;

#pragma unroll
    for (int s = 0; s < SPLIT_KEYS / 8; ++s) {
        // keep two steps in flight behind the one being consumed; vmcnt counts in issue order, so "all but the
        // youngest 2*n" retires step s (the q load above is older than every DMA that matters after step 0)
        if (s + 2 < SPLIT_KEYS / 8) { dma_step(s + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (s + 1 < SPLIT_KEYS / 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* slot = ring + (s % 3) * 2048;
        const bf16x8 kf = *(const bf16x8*)(slot + lane * 16);
        const bf16x8 vf = *(const bf16x8*)(slot + 1024 + lane * 16);
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) sc += qv[d] * bf2f((bf16_t)kf[d]);
        sc += __shfl_xor(sc, 1);
        sc += __shfl_xor(sc, 2);
        sc += __shfl_xor(sc, 4);
        if (k0 + s * 8 + kk >= Lk) sc = -INFINITY;
        const float m_new = fmaxf(sm, sc);
        if (m_new != -INFINITY) {
            const float alpha = (sm == -INFINITY) ? 0.f : ex2(sm - m_new);
            const float p = ex2(sc - m_new);
            const float pb = bf2f(f2bf(p));                 // P enters the PV product as bf16 (same rule as the MFMA path)
            sl = sl * alpha + p;
#pragma unroll
            for (int d = 0; d < 8; ++d) so[d] = so[d] * alpha + pb * bf2f((bf16_t)vf[d]);
            sm = m_new;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slot s read before step s+3 is issued into it (next iteration)
    }
    // merge the 8 key groups (lanes with equal sub), fixed tree
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(sm, off), l2 = __shfl_xor(sl, off);
        const float M = fmaxf(sm, m2);
        const float s1 = (sm == -INFINITY) ? 0.f : ex2(sm - M);
        const float s2 = (m2 == -INFINITY) ? 0.f : ex2(m2 - M);
        sl = sl * s1 + l2 * s2;
#pragma unroll
        for (int d = 0; d < 8; ++d) so[d] = so[d] * s1 + __shfl_xor(so[d], off) * s2;
        sm = M;
    }
    // chunk partial -> HBM (write-through): ml[unit][c] = (m, l), po[unit][c][64]
    float* po = a.part_o + ((size_t)unit * nc + c) * 64;
    float* pml = a.part_ml + ((size_t)unit * nc + c) * 2;
    if (kk == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) __hip_atomic_store(po + sub * 8 + d, so[d], RLX_AGENT);
        if (sub == 0) { __hip_atomic_store(pml, sm, RLX_AGENT); __hip_atomic_store(pml + 1, sl, RLX_AGENT); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(a.cnt + unit, 1u, RLX_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(nchunks - 1)) return;
    if (lane == 0) __hip_atomic_store(a.cnt + unit, 0u, RLX_AGENT);      // all chunks have arrived: ready for the next launch
    // the last chunk to arrive merges all chunks in chunk order; lane = output dim
    float M = -INFINITY, L = 0.f, O = 0.f;
    const float* bo = a.part_o + (size_t)unit * nc * 64 + lane;
    const float* bml = a.part_ml + (size_t)unit * nc * 2;
    for (int i0 = 0; i0 < nchunks; i0 += 8) {             // 8 chunks' partials requested before the first merge
        float pm[8], pl[8], pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = i0 + i < nchunks;
            const int ii = ok ? i0 + i : 0;
            pm[i] = __hip_atomic_load(bml + 2 * ii, RLX_AGENT);
            pl[i] = __hip_atomic_load(bml + 2 * ii + 1, RLX_AGENT);
            pv[i] = __hip_atomic_load(bo + (size_t)ii * 64, RLX_AGENT);
            if (!ok) { pm[i] = -INFINITY; pl[i] = 0.f; pv[i] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float Mn = fmaxf(M, pm[i]);
            const float s1 = (M == -INFINITY) ? 0.f : ex2(M - Mn);
            const float s2 = (pm[i] == -INFINITY) ? 0.f : ex2(pm[i] - Mn);
            L = L * s1 + pl[i] * s2;
            O = O * s1 + pv[i] * s2;
            M = Mn;
        }
    }
    a.ctx[(size_t)m * D + head * 64 + lane] = f2bf(O / L);
}

}  // namespace

hipError_t launch_attn_split(const TxtSplitArgs& a_in, hipStream_t s) {
    TxtSplitArgs a = a_in;
    const int M = a.rows * a.T;
    if (M <= 0 || a.H * 64 != a.D || a.beams <= 0) return hipErrorInvalidValue;
    a.nchunk_max = (a.S_img + a.t0 + a.T + SPLIT_KEYS - 1) / SPLIT_KEYS;
    hipLaunchKernelGGL(attn_split_kernel, dim3(M * a.H * a.nchunk_max), dim3(64), 0, s, a);
    return hipGetLastError();
}
int attn_split_chunks(int keys) { return (keys + SPLIT_KEYS - 1) / SPLIT_KEYS; }
