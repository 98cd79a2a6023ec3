# gemm256 of the current build against variant builds (gitcap/libgitcap_v*.so, built with an experiment macro) in ONE process: bitwise check,
# then interleaved per-launch times on the image-pass shapes.   python tools/gemm_variants_ab.py v1 v2
import sys, ctypes, torch
dev = torch.device('cuda:0')
G_ = 'real-time-video-captioning_amd/gitcap/'
names = ['base'] + sys.argv[1:]
libs = [ctypes.CDLL(G_ + 'libgitcap.so')] + [ctypes.CDLL(G_ + 'libgitcap_%s.so' % n) for n in sys.argv[1:]]
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def mk(M, N, K, epi):
    g = torch.Generator(device='cpu').manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); W = (torch.randn(N, K, generator=g) / K**0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev); resid = torch.randn(M, N, generator=g).to(dev) if epi == 3 else None
    return A, W, bias, resid
def run(lib, ops, M, N, K, epi, iters):
    A, W, bias, resid = ops
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    call = lambda: lib.gitcap_dbg_gemm(p(A), p(W), p(bias), p(resid), p(out), M, N, K, epi, 256, st())
    assert call() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out
shapes = [(18944, 2304, 768, 0), (18944, 3072, 768, 1), (18944, 3072, 768, 2), (18944, 768, 768, 0), (18944, 768, 3072, 0), (10496, 4096, 1024, 1)]
for M, N, K, epi in shapes:
    ops = mk(M, N, K, epi)
    ref = run(libs[0], ops, M, N, K, epi, 1)[1]
    same = [bool(torch.equal(run(l, ops, M, N, K, epi, 1)[1], ref)) for l in libs[1:]]
    for _ in range(3): [run(l, ops, M, N, K, epi, 10) for l in libs]
    r = []
    for rnd in range(6):                      # rotate who goes first
        order = list(range(len(libs)))[rnd % len(libs):] + list(range(len(libs)))[:rnd % len(libs)]
        t = [0.0] * len(libs)
        for i in order: t[i] = run(libs[i], ops, M, N, K, epi, 30)[0]
        r.append(t)
    med = [sorted(x[i] for x in r)[3] for i in range(len(libs))]
    print('M=%5d N=%4d K=%4d epi=%d  ' % (M, N, K, epi) + '  '.join('%s %.1f us%s' % (n, t, '' if i == 0 else ' (%+.1f %%%s)' % ((t / med[0] - 1) * 100, '' if same[i - 1] else ', DIFFERS'))
                                                                       for i, (n, t) in enumerate(zip(names, med))), flush=True)
