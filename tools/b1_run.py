# B clips of F frames (default: the webcam shape, one clip of 6 frames; B=32 F=1 = BASELINE configs[1]) under rocprofv3: `rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -- python3 tools/b1_run.py`
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
B = int(os.environ.get('B', '1')); L = int(os.environ.get('TOKENS', '20'))
F = int(os.environ.get('F', '6'))
cfg = git_base(F if F > 1 else 0); m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=B, max_frames=F, max_text_len=25, stop='never')
fr = torch.randn(B, F, 3, 224, 224, device='cuda')
for _ in range(int(os.environ.get('PASSES', '10'))): m.greedy_decode(fr, max_len=L)
torch.cuda.synchronize()
