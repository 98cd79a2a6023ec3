# BASELINE configs[4] (GIT-large, 10 frames, beam 4, 15 steps): synchronous gitcap_beam_search against the pipelined
# gitcap_beam_search_submit / _wait with 2 / 3 / 4 submissions in flight, interleaved rounds on one box.
#   B=4 STORAGE=fp8_e4m3 COMPUTE=fp8_ffn python tools/cfg4_pipeline.py        (PIPE=1 PASSES=n: only the pipelined loop, for rocprofv3)
import sys, os, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_large
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights, quantize_weights_fp8
cfg = git_large(10); B = int(os.environ.get('B', '4'))
wq = quantize_weights_fp8(synthetic_weights(cfg, 0))
m = GitCaptioner(cfg, wq, max_batch=B, max_frames=10, max_text_len=20, max_beams=4, weight_dtype=os.environ.get('STORAGE', 'fp8_e4m3'),
                 compute=os.environ.get('COMPUTE', 'fp8_ffn'))
g = torch.Generator().manual_seed(3)
ins = [torch.randn(B, 10, 3, 224, 224, generator=g).cuda() for _ in range(2)]
def pipe(n, depth):
    pend = []
    for i in range(n):
        pend.append(m.infer_async(ins[i % 2], beam_size=4, max_steps=15))
        if len(pend) == depth: pend.pop(0).result()
    while pend: pend.pop(0).result()
def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
if os.environ.get('PIPE'):
    pipe(int(os.environ.get('PASSES', '12')), 3); torch.cuda.synchronize(); sys.exit(0)
sync = lambda n: [m.infer(ins[i % 2], beam_size=4, max_steps=15) for i in range(n)]
sync(3); pipe(6, 3)
for rnd in range(3):
    r = {'sync': timed(sync, 8)}
    for d in (2, 3, 4):
        r['pipe%d' % d] = timed(lambda n: pipe(n, d), 12)
    print('B=%d %s/%s  ' % (B, os.environ.get('STORAGE', 'fp8_e4m3'), os.environ.get('COMPUTE', 'fp8_ffn')) +
          '  '.join('%s %.2f ms %.0f c/s' % (k, v, B * 1e3 / v) for k, v in r.items()), flush=True)
