# (needs tools/experiments/gemm256_s4_four_barriers.patch applied to csrc/gemm256.hip + the speed-switch lines of commit eba5b05: the product has neither)
# gemm256 with eight (product until round 5) vs four (S4, round 6) barriers per K-tile: bitwise equality on every epilogue incl. the
# fused GEMM + LayerNorm forms, then per-launch time on the image-pass shapes, interleaved in one process (speed switch 12), with
# torch.nn.functional.linear (hipBLASLt) on the same operands beside them.
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def mk(M, N, K, epi):
    g = torch.Generator(device='cpu').manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); W = (torch.randn(N, K, generator=g) / K**0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev); resid = torch.randn(M, N, generator=g).to(dev) if epi == 3 else None
    return A, W, bias, resid
def run(ops, M, N, K, epi, s4, iters=20):
    A, W, bias, resid = ops
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    lib.gitcap_dbg_config(12, s4)
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, 256, st())
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out
def run_ln(M, N, K, post, s4, iters=20):
    g = torch.Generator(device='cpu').manual_seed(M + N + K + post)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); W = (torch.randn(N, K, generator=g) / K**0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev); resid = (torch.randn(M, N, generator=g) * 2 + 0.5).to(dev)
    gamma, beta = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    of = torch.empty(M, N, device=dev); ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lib.gitcap_dbg_config(12, s4)
    call = lambda: lib.gitcap_dbg_gemm_ln(p(A), p(W), p(bias), p(resid), p(gamma), p(beta), ctypes.c_float(1e-5), p(of), p(ob), M, N, K, post, 1, 256, st())
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, of.clone(), ob.clone()
bad = []
for M, N, K, epi in [(256, 256, 64, 4), (512, 256, 128, 0), (1024, 768, 768, 0), (1280, 2304, 768, 1), (18944, 2304, 768, 0), (18944, 3072, 768, 2), (18944, 768, 768, 3),
                     (18944, 768, 3072, 4), (10496, 4096, 1024, 1), (768, 768, 192, 0)]:
    ops = mk(M, N, K, epi)
    ok = torch.equal(run(ops, M, N, K, epi, 0, 1)[1], run(ops, M, N, K, epi, 1, 1)[1])
    print('bitwise S4 vs 8-barrier M=%d N=%d K=%d epi=%d: %s' % (M, N, K, epi, ok), flush=True)
    if not ok: bad.append((M, N, K, epi))
for M, N, K, post in [(1024, 768, 768, 0), (18944, 768, 768, 0), (18944, 768, 3072, 1), (10496, 1024, 1024, 1), (75776, 768, 768, 0)]:
    a, b = run_ln(M, N, K, post, 0, 2), run_ln(M, N, K, post, 1, 2)
    ok = torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    print('bitwise S4 vs 8-barrier GEMM+LN M=%d N=%d K=%d post=%d: %s' % (M, N, K, post, ok), flush=True)
    if not ok: bad.append((M, N, K, 'ln', post))
assert not bad, bad
def ev(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(18944, 2304, 768, 0), (18944, 3072, 768, 0), (18944, 3072, 768, 1), (18944, 3072, 768, 2), (18944, 768, 768, 0), (18944, 768, 3072, 0), (18944, 768, 768, 3),
          (10496, 3072, 1024, 0), (10496, 4096, 1024, 1), (10496, 1024, 4096, 0)]
for M, N, K, epi in shapes:
    ops = mk(M, N, K, epi)
    for _ in range(4): run(ops, M, N, K, epi, 0)
    r = [[run(ops, M, N, K, epi, s4, 30)[0] for s4 in (0, 1)] for _ in range(4)]
    a, b = sorted(x[0] for x in r)[1], sorted(x[1] for x in r)[1]
    bb = ops[2].bfloat16()
    libt = ev(lambda: torch.nn.functional.linear(ops[0], ops[1], bb)) if epi == 0 else float('nan')
    print('M=%5d N=%4d K=%4d epi=%d   8-barrier %.1f us %.0f TF/s   S4 %.1f us %.0f TF/s   (%+.1f %%)   hipBLASLt %.1f us   rounds: %s'
          % (M, N, K, epi, a, 2.0 * M * N * K / a / 1e6, b, 2.0 * M * N * K / b / 1e6, (b / a - 1) * 100, libt,
             ' '.join('%.1f/%.1f' % (x[0], x[1]) for x in r)), flush=True)
for M, N, K, post in [(18944, 768, 768, 0), (18944, 768, 3072, 0), (18944, 768, 768, 1), (18944, 768, 3072, 1), (10496, 1024, 1024, 1)]:
    r = [[run_ln(M, N, K, post, s4, 30)[0] for s4 in (0, 1)] for _ in range(4)]
    a, b = sorted(x[0] for x in r)[1], sorted(x[1] for x in r)[1]
    print('GEMM+LN M=%5d N=%4d K=%4d post=%d   8-barrier %.1f us   S4 %.1f us   (%+.1f %%)' % (M, N, K, post, a, b, (b / a - 1) * 100), flush=True)
lib.gitcap_dbg_config(12, 0)
