import sys, ctypes, torch
dev = torch.device('cuda:0')
def bench(libpath, M, N, K, epi, tile=256, iters=20):
    lib = ctypes.CDLL(libpath)
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tile, st)
    assert call() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
libs = {'full': 'real-time-video-captioning_amd/gitcap/libgitcap.so', 'noDMA': 'tools/libgitcap_A.so', 'noLDSread': 'tools/libgitcap_B.so', 'neither': 'tools/libgitcap_C.so'}
for (N, K, epi) in [(768, 768, 0), (768, 3072, 0), (3072, 768, 0)]:
    print(N, K, epi, {k: round(bench(v, 18944, N, K, epi), 1) for k, v in libs.items()})
