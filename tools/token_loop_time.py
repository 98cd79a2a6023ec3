# GPU time of the greedy token loop alone: greedy_decode(max_len=L) - greedy_decode(max_len=1), HIP events, un-profiled.
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); w = synthetic_weights(cfg, 0)
L = int(os.environ.get('TOKENS', '20'))
for B in [int(b) for b in os.environ.get('BATCHES', '16,1').split(',')]:
    m = GitCaptioner(cfg, w, max_batch=B, max_frames=6, max_text_len=max(L, 25), stop='never')
    fr = torch.randn(B, 6, 3, 224, 224, device='cuda')
    def timed(ml, n=12):
        for _ in range(3): m.greedy_decode(fr, max_len=ml)
        torch.cuda.synchronize(); ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ids = m.greedy_decode(fr, max_len=ml); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort(); return ts[len(ts) // 2], int(ids.sum())
    t1, _ = timed(1); tL, chk = timed(L)
    print('B=%d F=6: max_len=1 %.3f ms, max_len=%d %.3f ms -> token loop %.3f ms = %.1f us/token  checksum %d'
          % (B, t1, L, tL, tL - t1, (tL - t1) * 1e3 / (L - 1), chk), flush=True)
    del m
