# Token loop of configs[2] (16 x 6-frame clips, 20 tokens) with kv_cache = bf16 / v_e4m3, interleaved: serial greedy(20) - greedy(1),
# and the txt_block launches alone (HIP-event brackets, class attn_text).
import sys, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); w = synthetic_weights(cfg, 0)
ms = {k: GitCaptioner(cfg, w, max_batch=16, max_frames=6, max_text_len=20, stop='never', kv_cache=k) for k in ('bf16', 'v_e4m3')}
fr = torch.randn(16, 6, 3, 224, 224, device='cuda')
def timed(m, ml, n=8):
    for _ in range(2): m.greedy_decode(fr, max_len=ml)
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.greedy_decode(fr, max_len=ml); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for rnd in range(3):
    for k, m in ms.items():
        t1, t20 = timed(m, 1), timed(m, 20)
        m.profile(True); m.greedy_decode(fr, max_len=20); torch.cuda.synchronize(); p = m.profile_read(); m.profile(False)
        print('round %d kv_cache=%-7s greedy(1) %.3f ms, greedy(20) %.3f ms -> token loop %.3f ms; txt_block %.2f us per launch (%d launches)'
              % (rnd, k, t1, t20, (t20 - t1) * 20 / 19, p['attn_text']['ms'] / p['attn_text']['launches'] * 1e3, p['attn_text']['launches']), flush=True)
