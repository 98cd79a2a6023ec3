# B=1 (webcam shape: one clip of 6 frames) under rocprofv3: `rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -- python3 tools/b1_run.py`
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
B = int(os.environ.get('B', '1')); L = int(os.environ.get('TOKENS', '20'))
cfg = git_base(6); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=B, max_frames=6, max_text_len=25, stop='never')
fr = torch.randn(B, 6, 3, 224, 224, device='cuda')
for _ in range(int(os.environ.get('PASSES', '10'))): m.greedy_decode(fr, max_len=L)
torch.cuda.synchronize()
