# Where does a host-fed batch lose time against the HBM-resident headline?  configs[2] (16 x 6 frames, GIT-base, 20 tokens), 3 in
# flight, variants: resident in / device out (the headline), resident in / CPU out (a host sync per result), page-locked and pageable
# host inputs through the staging ring with its own copy stream or on the caller's stream (GITCAP_COPY_STREAM).
import os, sys, time, torch
sys.path.insert(0, 'real-time-video-captioning_amd')
from gitcap.config import git_base
from gitcap.model import GitCaptioner
from gitcap.weights import synthetic_weights
dev = torch.device('cuda:0')
cfg = git_base(6)
w = synthetic_weights(cfg, 0)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (16, 6, 224, 224, 3), dtype=torch.uint8, generator=g) for _ in range(4)]
f32 = [torch.randn(16, 6, 3, 224, 224, generator=g) for _ in range(4)]
def region(m, ins, steps=24, inflight=3, to_cpu=False):
    pend = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        pend.append(m.greedy_decode_async(ins[i % 4], max_len=20, stop="never"))
        if len(pend) == inflight:
            r = pend.pop(0).result()
            if to_cpu: r = r.cpu()
    for f in pend:
        r = f.result()
        if to_cpu: r = r.cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
def med(m, ins, **kw):
    region(m, ins, steps=6, **kw)
    return sorted(region(m, ins, **kw) for _ in range(3))[1]
for mode in ("own", "caller"):
    os.environ["GITCAP_COPY_STREAM"] = mode
    m = GitCaptioner(cfg, w, device=dev, max_batch=16, max_frames=6, max_text_len=20, stop="never")
    res = {}
    if mode == "own":
        res["resident f32, device out"] = med(m, [x.to(dev) for x in f32])
        res["resident f32, CPU out"] = med(m, [x.to(dev) for x in f32], to_cpu=True)
        res["resident u8, device out"] = med(m, [x.to(dev) for x in u8])
    res["pinned u8"] = med(m, [x.pin_memory() for x in u8])
    res["pinned f32"] = med(m, [x.pin_memory() for x in f32])
    res["pageable u8"] = med(m, u8)
    res["pageable f32"] = med(m, f32)
    for k, v in res.items():
        print("copy stream %-6s  %-28s %.3f ms per batch = %.0f captions/s" % (mode, k, v, 16e3 / v), flush=True)
    # host cost of one staging step alone
    ring = m._staging()
    for name, x in (("pageable f32 57.8 MB", f32[0]), ("pageable u8 14.5 MB", u8[0])):
        t0 = time.perf_counter()
        for _ in range(5):
            dv, e, st = ring.stage(m, [x], x.dtype)
        torch.cuda.synchronize()
        print("  stage(%s): %.2f ms host + copy per call" % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    del m
