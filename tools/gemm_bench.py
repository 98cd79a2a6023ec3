# per-shape GEMM microbenchmark through the C-ABI debug hook (interleaved rounds in one process)
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
def run(M, N, K, epi, tile, iters=20):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tile, st)
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9, out, (A, W, bias, resid)
def check(M, N, K, epi, tile):
    ms, tf, out, (A, W, bias, resid) = run(M, N, K, epi, tile, iters=1)
    ref = A.float() @ W.float().t() + bias
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 2: ref = torch.nn.functional.gelu(ref)
    if epi == 3: ref = ref + resid
    err = (out.float() - ref).abs().max().item()
    return err
M = 18944
shapes = [(768, 768, 0), (768, 768, 3), (2304, 768, 0), (3072, 768, 1), (3072, 768, 2), (768, 3072, 3), (1536, 768, 0)]

print("check 256:", [round(check(18944, n, k, e, 256), 4) for n, k, e in [(768, 768, 3), (768, 3072, 3), (2304, 768, 0)]])
for rnd in range(2):
    for N, K, epi in shapes:
        r = []
        for tile in (128, 256):
            ms, tf, _, _ = run(M, N, K, epi, tile)
            r.append('%d: %.1f us %.0f TF/s' % (tile, ms * 1e3, tf))
        print('N=%4d K=%4d epi=%d  ' % (N, K, epi) + '   '.join(r))
