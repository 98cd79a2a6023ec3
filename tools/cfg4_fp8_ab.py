# BASELINE configs[4] (GIT-large, 4 clips x 10 frames, beam 4, 15 steps, e4m3 storage): bf16 compute vs compute="fp8_ffn"
# (FC1 / FC2 of the image rows on fp8 MFMA), interleaved on one box; per-kernel durations with ROCPROF=1 under rocprofv3.
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_large
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights, quantize_weights_fp8
cfg = git_large(10); B = 4
wq = quantize_weights_fp8(synthetic_weights(cfg, 0))
fr = torch.randn(B, 10, 3, 224, 224, device='cuda')
def med(fn, n=7):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]
ms = {c: GitCaptioner(cfg, wq, max_batch=B, max_frames=10, max_text_len=20, max_beams=4, weight_dtype='fp8_e4m3', compute=c) for c in ('bf16', 'fp8_ffn')}
for rnd in range(2):
    for c, m in ms.items():
        dt = med(lambda: m.infer(fr, beam_size=4, max_steps=15)); di = med(lambda: m.forward_image_enc(fr))
        print('round %d compute=%-7s: batch %.2f ms (%.1f captions/s), image pass %.2f ms, search loop %.2f ms' % (rnd, c, dt, B * 1e3 / dt, di, dt - di), flush=True)
a, b = (m.infer(fr, beam_size=4, max_steps=15)['predictions'] for m in ms.values())
print('captions equal between the two computes:', bool(torch.equal(a, b)))
