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
# configs[4]: GIT-large (ViT-L/14), e4m3 weight STORAGE (vs bf16 storage of the same values), 10-frame clips, beam=4, KV-cache decode
from gitcap.weights import quantize_weights_fp8
cfg = git_large(10); B = 4
wq = quantize_weights_fp8(synthetic_weights(cfg, 0))
fr = torch.randn(B, 10, 3, 224, 224, device='cuda')
GFLOP_PER_CAPTION = 1975.0            # SURVEY.md par. 8(d): GIT-large F=10, beam 4, 15 steps
for storage in ('fp8_e4m3', 'bf16'):
    m = GitCaptioner(cfg, wq, max_batch=B, max_frames=10, max_text_len=20, max_beams=4, weight_dtype=storage)
    wb = m.weight_bytes()
    if storage == 'fp8_e4m3':
        dt = timeit(lambda: m.infer(fr, beam_size=4, max_steps=15, on_device=False), n=3)
        print('configs[4] GIT-large e4m3 weights F=10 beam=4 15 steps, B=%d, host-side search loop: %.1f ms/batch  %.1f captions/s' % (B, dt * 1e3, B / dt))
    dt = timeit(lambda: m.infer(fr, beam_size=4, max_steps=15), n=3)
    di = timeit(lambda: m.forward_image_enc(fr), n=3)
    # search loop = 14 decoder steps over 16 rows: decoder + head weights once per step (e4m3: 1 B, bf16: 2 B per weight)
    # + the image K/V of the 4 clips for each of their 4 beams (the beams of a clip share it through L2) + text K/V
    D, V, Ld, S = cfg.dec_width, cfg.vocab_size, cfg.dec_layers, 10 * cfg.tokens_per_frame
    wbytes = (Ld * (4 * D * D + 2 * D * cfg.dec_ffn) + D * V) * (1.0 if storage == 'fp8_e4m3' else 2.0)
    kv = sum(B * 4 * Ld * 2 * (S + t + 1) * D * 2.0 for t in range(14))
    loop = dt - di
    print('configs[4] %s storage (%.0f MB of tensors on the device): device-resident search %.1f ms/batch  %.1f captions/s = %.3f of the '
          '2.5 PF MFMA peak; image pass (ViT-L/14 x %d frames + projection + decoder image prefix) %.1f ms; search loop %.1f ms = '
          '%.2f TB/s algorithmic = %.3f of the 8 TB/s HBM roofline' % (storage, wb / 1e6, dt * 1e3, B / dt, B / dt * GFLOP_PER_CAPTION / 2.5e6,
          B * 10, di * 1e3, loop * 1e3, (14 * wbytes + kv) / loop / 1e12, (14 * wbytes + kv) / loop / 8e12), flush=True)
    del m
