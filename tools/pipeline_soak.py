# Race screen for the pipelined path: N submissions over a few distinct inputs, up to 4 in flight, every result
# compared bit for bit with the synchronous greedy_decode of the same input.
import sys, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=16, max_frames=6, max_text_len=20, stop='never')
g = torch.Generator().manual_seed(5)
inputs = [torch.randn(b, 6, 3, 224, 224, generator=g).cuda() for b in (16, 16, 7, 16, 1)]
want = [m.greedy_decode(x, max_len=20) for x in inputs]
bad = 0; pend = []; N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(N):
    k = (i * 7 + i // 3) % len(inputs)
    pend.append((k, m.greedy_decode_async(inputs[k], max_len=20)))
    while len(pend) >= 4 - (i % 3 == 0):                 # keep 3 or 4 in flight (the library has four slots)
        k0, f = pend.pop(0); bad += int(not torch.equal(f.result(), want[k0]))
for k0, f in pend: bad += int(not torch.equal(f.result(), want[k0]))
print('pipeline soak: %d submissions, %d mismatches' % (N, bad)); sys.exit(1 if bad else 0)
