# Times the other BASELINE.json configs (parity-test cases, not the bench line): cfg1 and cfg4.
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_base, git_large
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# configs[1]: batch=32 single-frame image captioning, GIT-base bf16
cfg = git_base(0); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=32, max_frames=1, max_text_len=20)
fr = torch.randn(32, 1, 3, 224, 224, device='cuda')
dt = timeit(lambda: m.greedy_decode(fr, max_len=20, stop='never'))
print('configs[1] B=32 F=1 T=20 greedy serial: %.2f ms/batch  %.0f captions/s' % (dt * 1e3, 32 / dt))
pend = []
def pipe(n=12):
    global pend
    for _ in range(n):
        pend.append(m.greedy_decode_async(fr, max_len=20, stop='never'))
        if len(pend) == 4: pend.pop(0).result()
    while pend: pend.pop(0).result()
pipe(4); torch.cuda.synchronize(); t0 = time.perf_counter(); pipe(20); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print('configs[1] pipelined (4 in flight): %.2f ms/batch  %.0f captions/s' % (dt * 1e3, 32 / dt))
del m
# configs[4]: GIT-large (ViT-L/14), fp8 weight values, 10-frame clips, beam=4, KV-cache decode
cfg = git_large(10); B = 4
m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=B, max_frames=10, max_text_len=20, max_beams=4, weight_dtype='fp8_e4m3')
fr = torch.randn(B, 10, 3, 224, 224, device='cuda')
dt = timeit(lambda: m.infer(fr, beam_size=4, max_steps=15, on_device=False), n=3)
print('configs[4] GIT-large fp8-weights F=10 beam=4 15 steps, B=%d, host-side search loop: %.1f ms/batch  %.1f captions/s' % (B, dt * 1e3, B / dt))
dt = timeit(lambda: m.infer(fr, beam_size=4, max_steps=15), n=3)
print('configs[4] same, device-resident search: %.1f ms/batch  %.1f captions/s' % (dt * 1e3, B / dt))
dt = timeit(lambda: m.forward_image_enc(fr), n=3)
print('   of which image pass (ViT-L/14 x 40 frames + projection + decoder image prefix): %.1f ms' % (dt * 1e3))
