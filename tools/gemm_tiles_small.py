# small-M GEMM shapes (1, 2, 4, 8 clips of 6 frames) per tile kernel: where are the crossovers of launch_gemm_auto?
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
for B in (1, 2, 4, 8):
    M = (B * 6 * 197 + 255) // 256 * 256
    for N, K, epi in [(768, 768, 3), (768, 3072, 3), (2304, 768, 0), (3072, 768, 1)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        r = []
        for tile in (64, 128, 256):
            us = ev(lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tile, st))
            r.append('%d: %5.1f us' % (tile, us))
        print('B=%d M=%5d N=%4d K=%4d epi=%d  tiles128=%4d   %s' % (B, M, N, K, epi, (M // 128) * (N // 128), '   '.join(r)), flush=True)
