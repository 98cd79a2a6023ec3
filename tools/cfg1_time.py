# GPU time of greedy_decode for BASELINE configs[1] (32 single frames, 20 tokens) and of its token loop; HIP events, median of 12
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(0); w = synthetic_weights(cfg, 0)
m = GitCaptioner(cfg, w, max_batch=32, max_frames=1, max_text_len=25, stop='never')
fr = torch.randn(32, 1, 3, 224, 224, device='cuda')
def timed(ml, n=12):
    for _ in range(3): m.greedy_decode(fr, max_len=ml)
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.greedy_decode(fr, max_len=ml); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
t1 = timed(1); t20 = timed(20)
print('configs[1] B=32 F=1: max_len=1 %.3f ms, max_len=20 %.3f ms -> token loop %.3f ms = %.1f us/token' % (t1, t20, t20 - t1, (t20 - t1) * 1e3 / 19), flush=True)
