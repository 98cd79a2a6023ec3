# A/B of two libgitcap builds on the flash-attention shapes (+ bitwise comparison): python tools/attn_ab.py name=path ...
import sys, ctypes, torch
libs = {a.split('=')[0]: ctypes.CDLL(a.split('=')[1]) for a in sys.argv[1:]}
for G, S, H in [(96, 197, 12), (16, 1182, 12), (40, 257, 16), (7, 100, 12)]:
    W = H * 64
    qkv = (torch.randn(G * S, 3 * W, device='cuda') * 1.5).bfloat16()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs, res = {}, {k: [] for k in libs}
    for k, lib in libs.items():
        ctx = torch.zeros(G * S, W, device='cuda', dtype=torch.bfloat16)
        assert lib.gitcap_dbg_attn_full(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(ctx.data_ptr()), G, S, H, st) == 0
        torch.cuda.synchronize(); outs[k] = ctx
    for rnd in range(3):
        for k, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.gitcap_dbg_attn_full(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(outs[k].data_ptr()), G, S, H, st)
            e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
    ks = list(libs); same = all(torch.equal(outs[ks[0]], outs[k]) for k in ks[1:])
    q, kk, v = qkv.float().view(G, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(q @ kk.transpose(-1, -2) / 8.0, -1) @ v
    err = (outs[ks[-1]].float().view(G, S, H, 64).permute(0, 2, 1, 3) - ref).abs().max().item()
    print('G=%d S=%d H=%d' % (G, S, H), {k: '%.1f us' % min(v) for k, v in res.items()}, 'bitwise-equal' if same else 'DIFFER', 'max err vs fp32 %.4f' % err, flush=True)
