# diagnostic: where does the 224-row LayerNorm epilogue differ from the 256x256 kernel's?
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, post = 1792, 768, 768, 0
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.zeros(M + 16, K, device="cuda", dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda", generator=g); resid = torch.randn(M, N, device="cuda", generator=g) * 2 + 0.5
gamma, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
res = {}
for fused in (1, 224, 257):
    of = torch.full((M, N), float("nan"), device="cuda"); ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    rc = lib.gitcap_dbg_gemm_ln(p(A), p(W), p(bias), p(resid), p(gamma), p(beta), ctypes.c_float(1e-5), p(of), p(ob), M, N, K, post, fused, 256, st)
    torch.cuda.synchronize(); res[fused] = (of, ob, rc)
x = A[:M].float() @ W.float().t() + bias + resid
ln = torch.nn.functional.layer_norm(x, (N,), gamma, beta, 1e-5)
for fused in (1, 224, 257):
    of, ob, rc = res[fused]
    d = (ob.float() - res[1][1].float()).abs()
    bad = d > 0
    print('fused', fused, 'rc', rc, 'x equal', torch.equal(of, res[1][0]), 'ob mismatches', int(bad.sum()), 'max err vs torch', float((ob.float() - ln).abs().max()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('  rows', rows[:40].tolist(), '... n', len(rows)); print('  cols', cols[:40].tolist(), '... n', len(cols))
        r0 = int(rows[0]); print('  row', r0, 'bad cols', bad[r0].nonzero().flatten().tolist()[:80])
        print('  rows mod 224 hist', torch.bincount(rows % 224, minlength=224).nonzero().flatten().tolist()[:60])
