# latency of small batches (the webcam script's B=1 and a few more): GITCAP_GEMM_SMALL_TILES is read at library load
import sys, time, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
cfg = git_base(6); w = synthetic_weights(cfg, 0)
for B in (1, 2, 4, 8):
    m = GitCaptioner(cfg, w, max_batch=B, max_frames=6, max_text_len=25, stop='never')
    fr = torch.randn(B, 6, 3, 224, 224, device='cuda')
    for _ in range(3): m.greedy_decode(fr, max_len=20)
    torch.cuda.synchronize(); ts = []
    for _ in range(15):
        t0 = time.perf_counter(); ids = m.greedy_decode(fr, max_len=20); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); print('SMALL_TILES=%s B=%d F=6 20 tokens: p50 %.2f ms  checksum %d' % (os.environ.get('GITCAP_GEMM_SMALL_TILES', 'default'), B, ts[7] * 1e3, int(ids.sum())), flush=True)
    del m
