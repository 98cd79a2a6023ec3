# Race screen for the pipelined device beam search (gitcap_beam_search_submit / _wait, per-slot beam state): N submissions over a
# few distinct inputs, greedy and beam mixed, 3-4 in flight, every result compared bit for bit with the synchronous call.
#   python tools/beam_pipeline_soak.py [N]            GIT-base, 6-frame clips, beam 4, 12 steps / greedy 12 tokens
import sys, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=8, max_frames=6, max_text_len=12, max_beams=4, stop='never')
g = torch.Generator().manual_seed(9)
inputs = [torch.randn(b, 6, 3, 224, 224, generator=g).cuda() for b in (8, 8, 3, 8, 1)]
want_b = [m.infer(x, beam_size=4, max_steps=12) for x in inputs]
want_b = [(r['predictions'].clone(), r['logprobs'].clone()) for r in want_b]
want_g = [m.greedy_decode(x, max_len=12).clone() for x in inputs]
bad = 0; pend = []; N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def check(kind, k, f):
    r = f.result()
    if kind == 'g': return int(not torch.equal(r, want_g[k]))
    return int(not (torch.equal(r['predictions'], want_b[k][0]) and torch.equal(r['logprobs'], want_b[k][1])))
for i in range(N):
    k = (i * 7 + i // 3) % len(inputs)
    kind = 'g' if i % 5 == 2 else 'b'
    pend.append((kind, k, m.greedy_decode_async(inputs[k], max_len=12) if kind == 'g' else m.infer_async(inputs[k], beam_size=4, max_steps=12)))
    while len(pend) >= 4 - (i % 3 == 0):
        bad += check(*pend.pop(0))
for p in pend: bad += check(*p)
print('beam pipeline soak: %d submissions (every 5th greedy), %d mismatches' % (N, bad)); sys.exit(1 if bad else 0)
