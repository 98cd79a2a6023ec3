# real-time use case of the reference (src/real_time_inference.py): ONE clip of 6 frames, max_len 25
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=1, max_frames=6, max_text_len=25, stop='never')
for B in (1,):
    fr = torch.randn(B, 6, 3, 224, 224, device='cuda')
    for _ in range(3): m.greedy_decode(fr, max_len=25)
    torch.cuda.synchronize(); ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m.greedy_decode(fr, max_len=25); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); print('B=%d F=6 max_len=25: p50 %.2f ms  min %.2f ms' % (B, ts[10] * 1e3, ts[0] * 1e3))
    m.profile(True); m.greedy_decode(fr, max_len=25); torch.cuda.synchronize(); p = m.profile_read(); m.profile(False)
    print({k: (round(v['ms'], 2), v['launches']) for k, v in p.items()})
    frc = fr.cpu()
    t0 = time.perf_counter(); out = m.greedy_decode(frc, max_len=25); t1 = time.perf_counter()
    print('CPU tensor in -> CPU ids out (as real_time_inference.py:57-59): %.2f ms' % ((t1 - t0) * 1e3), tuple(out.shape), out.device)
