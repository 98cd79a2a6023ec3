# A/B of libgitcap builds / tile kernels on the encoder GEMM shapes: python tools/gemm_ab.py name=path[:tile] ...
# (tile: 128, 256 (default), 257 persistent, 258 two-workgroups-per-CU)
# (interleaved rounds in one process; first max-abs-error of every lib against fp32 torch)
import sys, ctypes, torch
dev = torch.device('cuda:0')
libs = {a.split('=')[0]: ctypes.CDLL(a.split('=')[1].split(':')[0]) for a in sys.argv[1:]}
tiles = {a.split('=')[0]: int((a.split('=')[1].split(':') + ['256'])[1]) for a in sys.argv[1:]}
M = 18944
shapes = [(768, 768, 0), (768, 768, 3), (2304, 768, 0), (3072, 768, 1), (768, 3072, 3), (1536, 768, 0)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def mk(N, K, epi):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 3 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    return A, W, bias, resid, out
def call(k, t, N, K, epi):
    A, W, bias, resid, out = t
    return libs[k].gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tiles[k], st)
for N, K, epi in shapes:
    t = mk(N, K, epi)
    A, W, bias, resid, out = t
    ref = A.float() @ W.float().t() + bias
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 3: ref = ref + resid
    errs, outs = {}, {}
    for k, lib in libs.items():
        out.zero_(); assert call(k, t, N, K, epi) == 0; torch.cuda.synchronize()
        errs[k] = round((out.float() - ref).abs().max().item(), 4); outs[k] = out.clone()
    ks = list(libs)
    same = all(torch.equal(outs[ks[0]], outs[k]) for k in ks[1:])
    res = {k: [] for k in libs}
    for rnd in range(3):
        for k, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call(k, t, N, K, epi)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
    print('N=%4d K=%4d epi=%d' % (N, K, epi), {k: '%.1f us' % min(v) for k, v in res.items()}, 'err', errs, 'bitwise-equal' if same else 'DIFFER', flush=True)
