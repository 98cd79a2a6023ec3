# scratch: first on-GPU parity probe for the tiny config
import sys, time, numpy as np, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_tiny
from gitcap.weights import synthetic_weights
from gitcap.model import GitCaptioner
from oracle.git_oracle import GitOracle, make_frames
for F in (2, 0):
    cfg = git_tiny(F); w = synthetic_weights(cfg, 0)
    ob = GitOracle(cfg, w, emulate_bf16=True); of = GitOracle(cfg, w)
    fr = make_frames(2, max(1, F), cfg.image_size, 1234)
    m = GitCaptioner(cfg, w, max_batch=4, max_text_len=16)
    _, vis = m.forward_image_enc(fr)
    torch.cuda.synchronize()
    vb = ob.encode_frames(fr); vf = of.encode_frames(fr)
    print('F', F, 'visual dev-vs-bf16oracle', (vis.cpu()-vb).abs().max().item(), 'dev-vs-fp32', (vis.cpu()-vf).abs().max().item(), 'oracle bf-vs-fp32', (vb-vf).abs().max().item())
    g = np.random.default_rng(7); ids = torch.from_numpy(g.integers(1, cfg.vocab_size, size=(2, 6))).long(); ids[:,0]=cfg.cls_token_id
    lg = m.forward_decoder(ids, vis).cpu()
    lb, _ = ob.forward_output_logits(fr, ids); lf, _ = of.forward_output_logits(fr, ids)
    print('   logits dev-vs-bf16oracle', (lg-lb).abs().max().item(), 'dev-vs-fp32', (lg-lf).abs().max().item(), 'oracle bf-vs-fp32', (lb-lf).abs().max().item(), 'std', lf.std().item())
    out = m.greedy_decode(fr, max_len=8, stop='never').cpu()
    print('   greedy dev', out.tolist()); print('   greedy orc', ob.greedy_decode(fr, 8, stop='never').tolist())
