# LayerNorm over image rows (fp32 in, bf16 out): GB/s for 1 / 2 / 4 rows per wave (GITCAP_LN_ROWS is read once
# per process, so run this once per setting)
import sys, ctypes, torch, os
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap import _lib
lib = _lib.load(); dev = torch.device('cuda:0')
for rows, D in [(18944, 768), (18944, 1024)]:
    x = torch.randn(rows, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    ob = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda: lib.gitcap_dbg_layernorm(p(x), p(g), p(b), 1e-5, rows, D, None, p(ob), st)
    assert call() == 0
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-5)
    err = (ob.float() - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 20)
    print('GITCAP_LN_ROWS=%s rows=%d D=%d: %.1f us  %.0f GB/s  max err %.4f  checksum %.6f' % (os.environ.get('GITCAP_LN_ROWS', 'auto'), rows, D, best * 1e3, rows * D * 6 / best / 1e6, err, ob.float().sum().item()), flush=True)
