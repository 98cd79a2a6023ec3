# One or two 6-frame clips, 20 greedy tokens, device resident: a speed switch of libgitcap on / off, interleaved
#   python tools/one_clip_ab.py <key> [value_off]
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
key = int(sys.argv[1]); off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=2, max_frames=6, max_text_len=20, stop='never')
for B in (1, 2):
    fr = torch.randn(B, 6, 3, 224, 224, device='cuda')
    def t(n=15):
        for _ in range(3): m.greedy_decode(fr, max_len=20)
        torch.cuda.synchronize(); ts = []
        for _ in range(n):
            t0 = time.perf_counter(); m.greedy_decode(fr, max_len=20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[n // 2]
    want = m.greedy_decode(fr, max_len=20).clone()
    rows = []
    for rnd in range(3):
        on = t(); old = lib.gitcap_dbg_config(key, off)
        same = bool(torch.equal(m.greedy_decode(fr, max_len=20), want)); o = t(); lib.gitcap_dbg_config(key, old)
        rows.append((on, o, same))
    print('B=%d switch %d: on / off ms per 20-token caption: %s' % (B, key, '  '.join('%.3f / %.3f%s' % (a, b, '' if s else ' DIFFER') for a, b, s in rows)), flush=True)
