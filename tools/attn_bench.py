import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load()
def run(G, S, H, iters=20):
    W = H * 64
    qkv = (torch.randn(G * S, 3 * W, device='cuda') * 1.5).bfloat16(); ctx = torch.zeros(G * S, W, device='cuda', dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.gitcap_dbg_attn_full(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(ctx.data_ptr()), G, S, H, st)
    assert call() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 4.0 * G * H * S * S * 64 / ms / 1e9
for G, S, H in [(96, 197, 12), (16, 1182, 12)]:
    us, tf = run(G, S, H)
    print('G=%d S=%d H=%d: %.1f us  %.0f TF/s' % (G, S, H, us, tf))
