# txt_block at the bench's byte count with different unit counts: would a two-way key split -- 384 half-units on
# 256 CUs instead of 192 units -- shorten the launch?  Proxy without writing the split kernel: the same 58 MB of image K/V per layer as
# 16 clips x 1182 keys (192 (row, head) units, 16-wave workgroups: the bench), as 32 clips x 591 keys (384 units of half the keys:
# what a two-way key split launches, minus its extra merge), in the 8-wave form (two units per CU) and in the 16-wave form, and
# 21 clips x 1182 keys (252 units: every CU busy).  HIP-event brackets of the library around every txt_block launch (class attn_text),
# teacher-forced single positions t = 0..9 through gitcap_text_forward.
import sys, ctypes, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
from gitcap import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
def run(B, F, w8):
    cfg = git_base(F)
    m = GitCaptioner(cfg, synthetic_weights(cfg, 0), device=dev, max_batch=B, max_frames=F, max_text_len=12, stop="never")
    g = torch.Generator().manual_seed(B + F)
    fr = torch.randn(B, F, 3, 224, 224, generator=g).to(dev)
    old = lib.gitcap_dbg_config(9, w8)
    try:
        m.greedy_decode(fr, max_len=10)                      # warm
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            m.profile(True)
            m.greedy_decode(fr, max_len=10)
            torch.cuda.synchronize()
            p = m.profile_read()
            m.profile(False)
            res.append(p["attn_text"]["ms"] / p["attn_text"]["launches"] * 1e3)
        us = sorted(res)[1]
    finally:
        lib.gitcap_dbg_config(9, old)
    S = F * cfg.tokens_per_frame
    mb = B * 12 * (S + 5) * 2 * 64 * 2 / 1e6
    units = B * 12
    form = "8-wave" if (w8 and units > 256) else "16-wave"
    cus = min(units, 256) if form == "16-wave" else min((units + 1) // 2, 256)
    print("B=%2d F=%d: %3d units x %4d keys, %-7s workgroups: %5.1f us per launch, %5.1f MB -> %.2f TB/s, %5.1f GB/s per CU if spread over %d CUs"
          % (B, F, units, S, form, us, mb, mb / us, mb / us * 1e3 / cus, cus), flush=True)
    del m
for B, F, w8 in [(16, 6, 1), (32, 3, 1), (32, 3, 0), (21, 6, 1), (8, 6, 1), (16, 3, 1)]:
    run(B, F, w8)
