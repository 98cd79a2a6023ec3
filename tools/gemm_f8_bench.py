# fp8 tile kernel (gemm_f8.hip) against the bf16 one (gemm256.hip) on the same shapes, through the C-ABI debug hooks,
# interleaved rounds in one process.  Shapes: BASELINE configs[4] (GIT-large, 10 240 image rows) and the headline.
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

def timeit(call, iters=30):
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def bf16(M, N, K, epi):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    return timeit(lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), None, p(out), M, N, K, epi, 256, st()))

def f8(M, N, K, epi):
    A = torch.randint(0, 120, (M, K), device=dev, dtype=torch.uint8); W = torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8)
    ws = torch.ones(N, device=dev); bias = torch.randn(N, device=dev)
    esz = {4: 4, 0: 2, 8: 1, 9: 1}[epi]
    out = torch.empty(M * N * esz, device=dev, dtype=torch.uint8)
    fn = lib.gitcap_dbg_gemm_f8
    return timeit(lambda: fn(p(A), p(W), p(ws), ctypes.c_float(1 / 16), p(bias), p(out), M, N, K, epi, ctypes.c_float(16.0), st()))

shapes = [(10240, 4096, 1024), (10240, 1024, 4096), (10240, 1024, 1024), (10240, 3072, 1024), (18944, 3072, 768), (18944, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split(',')) for s in sys.argv[1:]]
for rnd in range(2):
    for M, N, K in shapes:
        fl = 2.0 * M * N * K
        tb = bf16(M, N, K, 0); tb1 = bf16(M, N, K, 1)
        t0 = f8(M, N, K, 0); t8 = f8(M, N, K, 8)
        print('M=%5d N=%4d K=%4d  bf16 bias->bf16 %.1f us %.0f TF/s | qgelu->bf16 %.1f | fp8 bias->bf16 %.1f us %.0f TF/s | qgelu->e4m3 %.1f us'
              % (M, N, K, tb * 1e3, fl / tb / 1e9, tb1 * 1e3, t0 * 1e3, fl / t0 / 1e9, t8 * 1e3), flush=True)
