# Does a small-footprint kernel co-reside with the 256x256 GEMM (8 waves x 226 VGPRs: 48 VGPRs per SIMD lane stay free)
# or displace it?  Stream A: 60 GEMMs (N=2304, K=768).  Stream B: a continuous train of single-wave probe kernels
# (tools/probe/probe.hip) of equal work but different VGPR footprint (NV=4: 24, NV=8: 42, NV=16: 62 VGPRs).
import ctypes, sys, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load(); probe = ctypes.CDLL('tools/probe/libprobe.so')
dev = torch.device('cuda:0'); M, N, K = 18944, 2304, 768
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16(); bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
blocks, fpb = int(sys.argv[1]) if len(sys.argv) > 1 else 192, 16384
pin = torch.randn(blocks * fpb, device=dev); pout = torch.empty(blocks * 256, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
p = lambda t: ctypes.c_void_p(t.data_ptr())
def gemms(n):
    for _ in range(n): lib.gitcap_dbg_gemm(p(A), p(W), p(bias), None, p(out), M, N, K, 0, 256, ctypes.c_void_p(sa.cuda_stream))
def run(nv):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nprobe = 0
    if nv:
        eb0.record(sb)
        for _ in range(2500): probe.probe_launch(nv, p(pin), p(pout), blocks, fpb, ctypes.c_void_p(sb.cuda_stream))
        eb1.record(sb); nprobe = 2500
    e0.record(sa); gemms(60); e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 60 * 1e3, (eb0.elapsed_time(eb1) / nprobe * 1e3 if nv else 0.0)
gemms(5); torch.cuda.synchronize()
def run_custom(name, launch, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record(sb)
    for _ in range(n): launch()
    b1.record(sb)
    e0.record(sa); gemms(60); e1.record(sa); torch.cuda.synchronize()
    print('%-58s GEMM %.1f us each, other kernel %.2f us each' % (name, e0.elapsed_time(e1) / 60 * 1e3, b0.elapsed_time(b1) / n * 1e3), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == 'mech':
    big_in = torch.randn(3072 * 6144 + 4096, device=dev)    # sized for the largest probe below (6144 floats per wave)
    big_out = torch.empty(3072 * 256, device=dev); lk = torch.zeros(8192, device=dev)
    sbp = ctypes.c_void_p(sb.cuda_stream)
    run_custom('nothing on the other stream', lambda: None, 1)
    run_custom('bandwidth hog, co-resident (3072 waves x 19 KB, 42 VGPRs)', lambda: probe.probe_launch(8, p(big_in), p(big_out), 3072, 4608, sbp), 600)
    run_custom('bandwidth hog, 64 VGPRs (3072 waves x 19 KB)', lambda: probe.probe_launch(24, p(big_in), p(big_out), 3072, 6144, sbp), 600)
    for sl in (40, 80):
        run_custom('CU lock-out only: 192 x 1024-thread groups sleeping (%d)' % sl, lambda: probe.lock_launch(p(lk), 192, sl, sbp), 400)
    run_custom('nothing on the other stream', lambda: None, 1)
    sys.exit(0)
for nv in (0, 4, 8, 16, 24, 0, 8, 16):
    g, pr = run(nv)
    print('probe NV=%2d (%s): GEMM %.1f us each, probe %.2f us each' % (nv, {0: 'none', 4: '24 VGPRs', 8: '42 VGPRs', 16: '62 VGPRs', 24: '64 VGPRs'}[nv], g, pr), flush=True)
