# residual GEMM + LayerNorm: the 256x256 kernel's fused epilogue vs GEMM then row kernel, bench shapes (M = 19200 = B16 x F6 x 197 padded)
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def ev(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 19200
for N, K, post in [(768, 768, 0), (768, 3072, 0), (768, 768, 1), (768, 3072, 1), (1024, 1024, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev)
    g, b = torch.randn(N, device=dev), torch.randn(N, device=dev)
    of = torch.empty(M, N, device=dev); ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda fused: lib.gitcap_dbg_gemm_ln(p(A), p(W), p(bias), p(resid), p(g), p(b), ctypes.c_float(1e-5), p(of), p(ob), M, N, K, post, fused, 256, st)
    plain = ev(lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(of), M, N, K, 3, 256, st))
    ln = ev(lambda: lib.gitcap_dbg_layernorm(p(of), p(g), p(b), ctypes.c_float(1e-5), M, N, p(of) if post else None, p(ob), st))
    fused = ev(lambda: call(1))
    print('N=%4d K=%4d post=%d   resid GEMM %.1f us + LayerNorm %.1f us = %.1f   fused %.1f us' % (N, K, post, plain, ln, plain + ln, fused), flush=True)
