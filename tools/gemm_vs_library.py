# Yardstick only (never in the product): the image-pass GEMM shapes through libgitcap's 256x256 kernel and through
# torch.nn.functional.linear (hipBLASLt/rocBLAS on ROCm) on the same operands: how far from the vendor library's best?
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
def ev(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 18944
for N, K, epi in [(768, 768, 0), (768, 768, 3), (2304, 768, 0), (3072, 768, 0), (3072, 768, 1), (3072, 768, 2), (768, 3072, 3), (768, 3072, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    ours = ev(lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, 256, st))
    bb = bias.bfloat16()
    libt = ev(lambda: torch.nn.functional.linear(A, W, bb))           # bf16 out, bias only: the library's plain GEMM
    mm = ev(lambda: torch.mm(A, W.t()))
    fl = 2.0 * M * N * K
    print('N=%4d K=%4d epi=%d   gemm256 %.1f us %.0f TF/s   F.linear(bf16 out) %.1f us %.0f TF/s   mm %.1f us %.0f TF/s'
          % (N, K, epi, ours, fl / ours / 1e6, libt, fl / libt / 1e6, mm, fl / mm / 1e6), flush=True)
