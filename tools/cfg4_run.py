# BASELINE configs[4] (GIT-large, 10 frames, beam 4, 15 steps, B=4, e4m3 storage) under rocprofv3:
#   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg4 --output-format csv -- python3 tools/cfg4_run.py
import sys, os, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.config import git_large
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights, quantize_weights_fp8
cfg = git_large(10); B = int(os.environ.get('B', '4'))
wq = quantize_weights_fp8(synthetic_weights(cfg, 0))
m = GitCaptioner(cfg, wq, max_batch=B, max_frames=10, max_text_len=20, max_beams=4, weight_dtype=os.environ.get('STORAGE', 'fp8_e4m3'), compute=os.environ.get('COMPUTE', 'bf16'))
fr = torch.randn(B, 10, 3, 224, 224, device='cuda')
for _ in range(int(os.environ.get('PASSES', '5'))): m.infer(fr, beam_size=4, max_steps=15)
torch.cuda.synchronize()
