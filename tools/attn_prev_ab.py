# attn_full of the current build against gitcap/libgitcap_prev.so (the previous build, see tools/ab_prev_so.sh) in ONE process:
# bitwise equality of ctx on the product shapes (+ ragged / half-tile / exact-multiple lengths), then interleaved per-launch times.
import ctypes, torch
dev = torch.device('cuda:0')
G_ = 'real-time-video-captioning_amd/gitcap/'
new, old = ctypes.CDLL(G_ + 'libgitcap.so'), ctypes.CDLL(G_ + 'libgitcap_prev.so')
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def mk(G, S, H, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed + S)
    return (torch.randn(G * S, 3 * H * 64, generator=g) * 1.5).to(dev).bfloat16()
def run(lib, qkv, G, S, H, iters):
    ctx = torch.zeros(G * S, H * 64, device=dev, dtype=torch.bfloat16)
    call = lambda: lib.gitcap_dbg_attn_full(p(qkv), p(ctx), G, S, H, st())
    assert call() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, ctx
bad = []
for G, S, H in [(96, 197, 12), (16, 1182, 12), (3, 64, 2), (2, 128, 1), (2, 65, 3), (1, 257, 16), (2, 96, 2), (1, 97, 1), (40, 257, 16), (4, 2570, 12), (5, 17, 2)]:
    qkv = mk(G, S, H)
    ok = torch.equal(run(new, qkv, G, S, H, 1)[1], run(old, qkv, G, S, H, 1)[1])
    print('bitwise new vs prev G=%d S=%d H=%d: %s' % (G, S, H, ok), flush=True)
    if not ok: bad.append((G, S, H))
assert not bad, bad
for G, S, H in [(96, 197, 12), (16, 1182, 12), (40, 257, 16), (4, 2570, 12)]:
    qkv = mk(G, S, H)
    for _ in range(3): run(new, qkv, G, S, H, 10); run(old, qkv, G, S, H, 10)
    r = [(run(new, qkv, G, S, H, 30)[0], run(old, qkv, G, S, H, 30)[0]) for _ in range(5)]
    a, b = sorted(x[0] for x in r)[2], sorted(x[1] for x in r)[2]
    fl = 4.0 * G * H * S * S * 64
    print('G=%3d S=%4d H=%2d   new %.1f us %.0f TF/s   prev %.1f us %.0f TF/s   (%+.1f %%)   rounds %s'
          % (G, S, H, a, fl / a / 1e6, b, fl / b / 1e6, (a / b - 1) * 100, ' '.join('%.1f/%.1f' % x for x in r)), flush=True)
