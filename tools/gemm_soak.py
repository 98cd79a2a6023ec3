# race screen for the big-tile GEMMs -- gemm256 (argv[1] = 224 / 257 selected the retired gemm_mt.hip: tools/experiments/)
# (hand-placed counted vmcnt / staggered barriers): many launches per shape, output
# poisoned with NaN before every launch, every result compared with the fp32 reference; second pass with a
# concurrent stream hammering HBM so that DMA arrival times vary
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def soak(M, N, K, epi, tile, iters, noise):
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K + epi)
    if tile == 224: M = (M + 223) // 224 * 224
    A = torch.zeros(M + 16, K, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device=dev, generator=g).bfloat16(); A = A[:M]; W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev, generator=g); resid = torch.randn(M, N, device=dev, generator=g) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    ref = A.float() @ W.float().t() + bias
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 3: ref = ref + resid
    side = torch.cuda.Stream(); big = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    worst, bad = 0.0, 0
    for it in range(iters):
        out.fill_(float('nan'))
        if noise:
            with torch.cuda.stream(side):
                big2.copy_(big, non_blocking=True)
        assert lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tile, st) == 0
        d = (out.float() - ref)
        e = float(d.abs().max()) if not torch.isnan(d).any() else float('inf')
        tol = 0.05 if out.dtype == torch.bfloat16 else 2e-3
        bad += e > tol; worst = max(worst, e)
    torch.cuda.synchronize()
    return worst, bad
for noise in (False, True):
    for (M, N, K, epi) in [(18944, 768, 768, 3), (18944, 2304, 768, 0), (18944, 3072, 768, 1), (18944, 768, 3072, 3), (2560, 1536, 768, 0), (256, 256, 64, 4), (512, 256, 128, 4)]:
        w, b = soak(M, N, K, epi, TILE, 150 if M > 4000 else 400, noise)
        print('noise=%d M=%d N=%d K=%d epi=%d: worst err %.4g, bad launches %d' % (noise, M, N, K, epi, w, b), flush=True)
