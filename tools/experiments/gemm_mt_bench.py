# 224-row tiles (gemm_mt.hip) vs the 256x256 kernel at the bench shape (18 912 valid rows), interleaved rounds in one process:
#   tile 256 -> M = 18944 (74 row blocks), 224 -> M = 19040 (85 row blocks), 257 = gemm_mt.hip on 256 rows.
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 18912
MA = (ROWS + 255) // 256 * 256 + 256
def ev(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def m_of(tile): return (ROWS + 223) // 224 * 224 if tile == 224 else (ROWS + 255) // 256 * 256
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(768, 768, 0), (2304, 768, 0), (3072, 768, 1), (3072, 768, 2), (768, 3072, 3), (768, 768, 3), (1536, 768, 0)]
for N, K, epi in shapes:
    A = torch.randn(MA, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(MA, N, device=dev) if epi == 3 else None
    out = torch.empty(MA, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    res = {}
    for rnd in range(3):
        for tile in (256, 224, 257):
            us = ev(lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), m_of(tile), N, K, epi, tile, st))
            res.setdefault(tile, []).append(us)
    print('plain  N=%4d K=%4d epi=%d  ' % (N, K, epi) + '   '.join('%d: %.1f us (%.0f TF/s alg)' % (t, min(v), 2.0 * ROWS * N * K / min(v) / 1e6) for t, v in res.items()), flush=True)
for N, K, post in [(768, 768, 0), (768, 3072, 0), (768, 768, 1), (768, 3072, 1)]:
    A = torch.randn(MA, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(MA, N, device=dev)
    g, b = torch.randn(N, device=dev), torch.randn(N, device=dev)
    of = torch.empty(MA, N, device=dev); ob = torch.empty(MA, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for fused in (1, 224, 257):
            us = ev(lambda: lib.gitcap_dbg_gemm_ln(p(A), p(W), p(bias), p(resid), p(g), p(b), ctypes.c_float(1e-5), p(of), p(ob), m_of(fused), N, K, post, fused, 256, st))
            res.setdefault(fused, []).append(us)
    print('+LN    N=%4d K=%4d post=%d ' % (N, K, post) + '   '.join('%d: %.1f us' % (t, min(v)) for t, v in res.items()), flush=True)
