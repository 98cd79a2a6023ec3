# diagnostic build only (tools/libgitcap_diag.so): per-workgroup s_memtime stamps of gemm256
import os, ctypes, torch, numpy as np
dev = torch.device('cuda:0')
M, K = 18944, 768
for N, epi in [(768, 0), (768, 3), (2304, 0), (3072, 1)]:
    ntiles = (M // 256) * (N // 256)
    dbg = torch.zeros(ntiles * 8 * 5, dtype=torch.int64, device=dev)
    os.environ['GEMM_DBG_PTR'] = str(dbg.data_ptr())
    lib = ctypes.CDLL('tools/libgitcap_diag.so')
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, 256, st)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(ntiles, 8, 5).astype(np.float64)
    t0, t1, t2, t3 = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    start = t0.min()
    print('N=%d epi=%d tiles=%d | per-wave cycles: prologue %.0f  loop %.0f  epilogue %.0f | kernel span %.0f cycles; block start spread: p50 %.0f p90 %.0f max %.0f; block end p50 %.0f max %.0f' % (
        N, epi, ntiles, np.median(t1 - t0), np.median(t2 - t1), np.median(t3 - t2), t3.max() - start,
        np.percentile(t0.min(1) - start, 50), np.percentile(t0.min(1) - start, 90), (t0.min(1) - start).max(), np.median(t3.max(1) - start), (t3.max(1) - start).max()))
