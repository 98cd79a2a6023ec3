# student decoder (SURVEY par. 8 f.2): latency of greedy_decode from memory, the webcam script's shape
# (src/real_time_inference.py:58: one 6-frame clip, max_len=25) and a few batch sizes
import sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd'); sys.path.insert(0, '.')
from gitcap.student import StudentCaptioner
from gitcap.student_config import student_base, student_synthetic_weights
cfg = student_base()
m = StudentCaptioner(cfg=cfg, weights=student_synthetic_weights(cfg, 0), max_batch=64, max_text_len=25)
for B in (1, 4, 16, 64):
    mem = torch.randn(B, 6, cfg.d_model, device='cuda'); ids = torch.empty(B, 26, dtype=torch.int64, device='cuda')
    for _ in range(3): m.greedy_decode(mem, max_len=25, stop='never')
    torch.cuda.synchronize(); t = []
    for _ in range(20):
        t0 = time.perf_counter(); m.greedy_decode(mem, max_len=25, stop='never'); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    t.sort()
    host = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m._call('gitcap_student_greedy', __import__('ctypes').c_void_p(mem.data_ptr()), B, 25, 0, __import__('ctypes').c_void_p(ids.data_ptr()), None, m._stream())
        host.append(time.perf_counter() - t0)
    host.sort()
    print('B=%2d  25-token greedy from memory: p50 %.2f ms  (%.0f captions/s, %.1f us per token step); host time to enqueue %.2f ms' % (B, t[10] * 1e3, B / t[10], t[10] / 25 * 1e6, host[5] * 1e3), flush=True)
