# host-side cost of enqueueing one batch (C launch loop + Python), vs the GPU step time
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=16, max_frames=6, max_text_len=20, stop='never')
fr = torch.randn(16, 6, 3, 224, 224, device='cuda')
for _ in range(3): m.greedy_decode(fr, max_len=20)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = m.greedy_decode_async(fr, max_len=20); t1 = time.perf_counter()
    f.result(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('host enqueue ms: %.2f   total (serial) ms: %.2f' % (1e3 * sorted(t[0] for t in ts)[5], 1e3 * sorted(t[1] for t in ts)[5]))
