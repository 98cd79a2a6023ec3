# one clip (the webcam case of src/real_time_inference.py: B=1, 6 frames, max_len 25) under rocprofv3:
#   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 --output-format csv -- python3 tools/b1_run.py
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
F = int(os.environ.get('F', '6')); cfg = git_base(F if F > 1 else 0); B = int(os.environ.get('B', '1')); T = int(os.environ.get('T', '25'))
m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=B, max_frames=F, max_text_len=T, stop='never')
fr = torch.randn(B, F, 3, 224, 224, device='cuda')
for _ in range(int(os.environ.get('PASSES', '10'))): m.greedy_decode(fr, max_len=T)
torch.cuda.synchronize()
