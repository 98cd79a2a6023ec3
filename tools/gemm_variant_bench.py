# gemm2w (tools/experiments/gemm2w.hip: 256 x 128 tiles, two workgroups per CU) against gemm256 per launch: bitwise equality and
# time, image-pass shapes.  Needs the diagnostic build: python tools/build_diag.py (tile code 258 lives only there)
import sys, ctypes, torch
lib = ctypes.CDLL('tools/libgitcap_diag.so')
dev = torch.device('cuda:0'); TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 258
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def mk(M, N, K, epi):
    g = torch.Generator(device='cpu').manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); W = (torch.randn(N, K, generator=g) / K**0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev); resid = torch.randn(M, N, generator=g).to(dev) if epi == 3 else None
    return A, W, bias, resid
def run(ops, M, N, K, epi, tile, iters=20):
    A, W, bias, resid = ops
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, tile, st())
    assert call() == 0, (M, N, K, epi, tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out
bad = []
for M, N, K, epi in [(1024, 768, 768, 0), (1280, 2304, 768, 0), (18944, 2304, 768, 0), (18944, 3072, 768, 1), (18944, 3072, 768, 2), (18944, 768, 768, 3),
                     (18944, 768, 3072, 4), (10496, 4096, 1024, 1)]:
    ops = mk(M, N, K, epi)
    ref = run(ops, M, N, K, epi, 256 if M % 256 == 0 else 128, 1)[1]
    got = run(ops, M, N, K, epi, TILE, 1)[1]
    ok = torch.equal(ref, got)
    print('bitwise vs gemm256 M=%d N=%d K=%d epi=%d: %s' % (M, N, K, epi, ok), flush=True)
    if not ok: bad.append((M, N, K, epi))
assert not bad, bad
shapes = [(18944, 2304, 768, 0), (18944, 3072, 768, 1), (18944, 3072, 768, 2), (18944, 768, 768, 0), (18944, 768, 3072, 0),
          (10496, 3072, 1024, 0), (10496, 4096, 1024, 1), (10496, 1024, 4096, 0)]
# devices differ by up to 12 % and clocks move with load: warm up, then interleave the two kernels on the same operands
for M, N, K, epi in shapes:
    ops = mk(M, N, K, epi)
    for _ in range(4): run(ops, M, N, K, epi, 256)
    r = [[run(ops, M, N, K, epi, tile, 30)[0] for tile in (256, TILE)] for _ in range(4)]
    a, b = sorted(x[0] for x in r)[1], sorted(x[1] for x in r)[1]
    print('M=%5d N=%4d K=%4d epi=%d   gemm256 %.1f us %.0f TF/s   variant %.1f us %.0f TF/s   (%+.1f %%)   rounds: %s'
          % (M, N, K, epi, a, 2.0 * M * N * K / a / 1e6, b, 2.0 * M * N * K / b / 1e6, (b / a - 1) * 100,
             ' '.join('%.1f/%.1f' % (x[0], x[1]) for x in r)), flush=True)
