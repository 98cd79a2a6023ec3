# attn_full of the current build against variant builds (gitcap/libgitcap_<name>.so) in ONE process: error against the fp64 reference of
# the same bf16 operands, then interleaved per-launch times with the order rotated.   python tools/attn_variants_ab.py poly4 poly8 poly16
import sys, ctypes, torch
dev = torch.device('cuda:0')
G_ = 'real-time-video-captioning_amd/gitcap/'
names = ['base'] + sys.argv[1:]
libs = [ctypes.CDLL(G_ + 'libgitcap.so')] + [ctypes.CDLL(G_ + 'libgitcap_%s.so' % n) for n in sys.argv[1:]]
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def run(lib, qkv, G, S, H, iters):
    ctx = torch.zeros(G * S, H * 64, device=dev, dtype=torch.bfloat16)
    call = lambda: lib.gitcap_dbg_attn_full(p(qkv), p(ctx), G, S, H, st())
    assert call() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, ctx
for G, S, H in [(4, 197, 12), (2, 1182, 12)]:
    g = torch.Generator(device='cpu').manual_seed(S)
    qkv = (torch.randn(G * S, 3 * H * 64, generator=g) * 1.5).to(dev).bfloat16()
    q, k, v = (t.double().view(G, S, H, 64).transpose(1, 2) for t in qkv.split(H * 64, dim=1))
    s = q @ k.transpose(-1, -2) * 0.125
    pr = torch.exp(s - s.max(-1, keepdim=True).values)
    ref = ((pr.bfloat16().double() @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(G * S, H * 64)
    print('G=%d S=%d: max |ctx - fp64 reference|: ' % (G, S) + '  '.join('%s %.4f' % (n, float((run(l, qkv, G, S, H, 1)[1].double() - ref).abs().max())) for n, l in zip(names, libs)), flush=True)
for G, S, H in [(96, 197, 12), (16, 1182, 12)]:
    g = torch.Generator(device='cpu').manual_seed(S)
    qkv = (torch.randn(G * S, 3 * H * 64, generator=g) * 1.5).to(dev).bfloat16()
    for _ in range(3): [run(l, qkv, G, S, H, 10) for l in libs]
    r = []
    for rnd in range(2 * len(libs)):
        order = list(range(len(libs)))[rnd % len(libs):] + list(range(len(libs)))[:rnd % len(libs)]
        t = [0.0] * len(libs)
        for i in order: t[i] = run(libs[i], qkv, G, S, H, 30)[0]
        r.append(t)
    med = [sorted(x[i] for x in r)[len(r) // 2] for i in range(len(libs))]
    print('G=%3d S=%4d H=%2d  ' % (G, S, H) + '  '.join('%s %.1f us%s' % (n, t, '' if i == 0 else ' (%+.1f %%)' % ((t / med[0] - 1) * 100)) for i, (n, t) in enumerate(zip(names, med))), flush=True)
