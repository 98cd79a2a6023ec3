# diagnostic build only (tools/libgitcap_diag.so = csrc with s_memtime stamps + HW_ID in gemm256):
# per-CU timeline of one launch: where a round's time goes (launch gap, prologue, K loop, epilogue)
import os, ctypes, torch, numpy as np
dev = torch.device('cuda:0')
M = 18944
for N, K, epi in [(768, 768, 0), (768, 768, 3), (2304, 768, 0), (3072, 768, 1), (768, 3072, 3)]:
    ntiles = (M // 256) * (N // 256)
    dbg = torch.zeros(ntiles * 8 * 8, dtype=torch.int64, device=dev)
    os.environ['GEMM_DBG_PTR'] = str(dbg.data_ptr())
    lib = ctypes.CDLL('tools/libgitcap_diag.so')
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, 256, st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    d = dbg.cpu().numpy().reshape(ntiles, 8, 8)
    t = d[..., :4].astype(np.float64)
    b0, b1, b2, b3 = t[..., 0].min(1), np.median(t[..., 1], 1), np.median(t[..., 2], 1), t[..., 3].max(1)
    xcc = d[:, 0, 5] & 0xf
    cu = (xcc << 8) | ((d[:, 0, 4] >> 8) & 0xff)
    r0, r3 = d[..., 6].min(1).astype(np.float64), d[..., 7].max(1).astype(np.float64)   # s_memrealtime: 100 MHz, global
    cyc_per_us = np.median((b3 - b0) / ((r3 - r0) / 100.0))                             # s_memtime ticks per us
    # s_memtime is not comparable between CUs: rebase every block on the global clock
    shift = r0 / 100.0 * cyc_per_us - b0
    b0, b1, b2, b3 = b0 + shift, b1 + shift, b2 + shift, b3 + shift
    start = b0.min(); span = b3.max() - start
    gaps, first, nblk = [], [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]; idx = idx[np.argsort(b0[idx])]
        first.append(b0[idx[0]] - start); nblk.append(len(idx))
        gaps += list(b0[idx[1:]] - b3[idx[:-1]])
    q = lambda x: ' '.join('%.2f' % (v / cyc_per_us) for v in np.percentile(x, [50, 90, 100])) if len(x) else '-'
    busy = (b2 - b1).sum() / (span * 256)
    print('N=%d K=%d epi=%d tiles=%d event %.1f us, stamped span %.1f us (%.0f ticks/us) CUs=%d blocks/CU %d..%d' % (N, K, epi, ntiles, us, span / cyc_per_us, cyc_per_us, len(np.unique(cu)), min(nblk), max(nblk)))
    print('   us p50/p90/max: first-start %s | same-CU gap %s | prologue %s | loop %s | epilogue %s | loop share of 256 CU x span: %.2f' % (
        q(first), q(gaps), q(b1 - b0), q(b2 - b1), q(b3 - b2), busy), flush=True)
