"""Host-fed submissions: the reference's callers hold their frames in HOST memory -- OpenCV uint8 frames in
src/real_time_inference.py:39-58, a DataLoader's CPU tensor in src/inference.py:45-51 (src/utils/dataloader.py:18-32, :60-82) --
and hand them to ``greedy_decode`` / the teacher's ``forward``.  Here they go through gitcap_greedy_submit / gitcap_greedy_raw_submit /
gitcap_beam_search_raw_submit (include/gitcap.h) behind a pinned staging ring and a copy stream (gitcap/model.py: _StagingRing).
Everything must be BITWISE what the device-resident synchronous calls return for the same frames."""
import numpy as np
import pytest
import torch

from gitcap.config import git_base, git_tiny
from gitcap.weights import synthetic_weights
from oracle.git_oracle import make_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def captioner_cls():
    from gitcap.model import GitCaptioner
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return GitCaptioner


def _camera(b, f, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (b, f, h, w, 3), dtype=torch.uint8, generator=g)


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_host_fed_greedy_bitwise(captioner_cls, size):
    """fp32 NCHW frames (pageable and page-locked) and uint8 camera frames (224 x 224: identity resize; 240 x 320: bicubic resize +
    crop) from host memory, mixed with device-resident submissions, ragged batches, more submissions than ring entries: every
    result equals the synchronous device-resident call's, and the raw form equals gitcap_preprocess followed by gitcap_greedy."""
    cfg = git_tiny(2) if size == "tiny" else git_base(2)
    S = cfg.image_size
    m = captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=4, max_frames=2, max_text_len=10, stop="never")
    f32 = [make_frames(b, 2, S, 900 + i) for i, b in enumerate((4, 3, 1))]
    cams = [_camera(4, 2, S, S, 1), _camera(2, 2, 240, 320, 2), _camera(3, 2, 300, 224, 3)]
    cases = []
    for x in f32:
        cases += [("f32 pageable", x), ("f32 pinned", x.clone().pin_memory()), ("f32 device", x.cuda())]
    for x in cams:
        cases += [("u8 pageable", x), ("u8 pinned", x.clone().pin_memory()), ("u8 device", x.cuda())]
    want = []
    for _, x in cases:
        want.append(m.greedy_decode(x.cuda(), max_len=9).cpu().clone())
    # the raw transform on the device == gitcap_preprocess + the fp32 path (what f.1 promises), also for the submissions
    from gitcap.preprocess import preprocess_frames
    for x, k in ((cams[1], 3 * len(f32) + 3), (cams[2], 3 * len(f32) + 6)):
        pre = preprocess_frames(x, S)
        assert torch.equal(m.greedy_decode(pre, max_len=9).cpu(), want[k])
    torch.cuda.synchronize()
    order = [(i * 7 + i // 4) % len(cases) for i in range(30)]
    pend, bad = [], []
    for n, k in enumerate(order):
        pend.append((k, m.greedy_decode_async(cases[k][1], max_len=9)))
        while len(pend) >= 4 - (n % 3 == 0):
            k0, f = pend.pop(0)
            r = f.result()
            assert r.device == cases[k0][1].device
            if not torch.equal(r.cpu(), want[k0]):
                bad.append((n, cases[k0][0], k0))
    for k0, f in pend:
        if not torch.equal(f.result().cpu(), want[k0]):
            bad.append(("tail", cases[k0][0], k0))
    assert not bad, bad
    # coalesced host-fed groups: two caller batches staged into one ring entry, one pass
    m2 = captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=8, max_frames=2, max_text_len=10, stop="never")
    futs = [m2.greedy_decode_async(f32[0] if i % 2 == 0 else f32[0].flip(0), max_len=9, coalesce=2) for i in range(6)]
    for i, f in enumerate(futs):
        w = want[0] if i % 2 == 0 else want[0].flip(0)
        assert torch.equal(f.result(), w), i
    futs = [m2.greedy_decode_async(cams[0], max_len=9, coalesce=2) for _ in range(3)]     # the last group stays half filled
    for f in futs:
        assert torch.equal(f.result(), want[3 * len(f32)])
    # the synchronous call with a CPU tensor (real_time_inference.py:57-58) goes through the staging entry of its own
    for k in (0, 1, 3 * len(f32), 3 * len(f32) + 3):
        r = m.greedy_decode(cases[k][1], max_len=9)
        assert r.device.type == "cpu" and torch.equal(r, want[k])
    big = torch.cat([f32[0], f32[1], f32[2]], 0)                      # 8 clips through max_batch 4: staged chunk by chunk
    assert torch.equal(m.greedy_decode(big, max_len=9), torch.cat([want[0], want[3], want[6]], 0))


def test_host_fed_beam_search_and_teacher_forward(captioner_cls):
    """infer_async / teacher_forward from host memory (fp32 and raw uint8 frames): predictions, log-probabilities, per-step logits
    and visual features bitwise those of the device-resident calls."""
    cfg = git_tiny(2)
    S = cfg.image_size
    m = captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=4, max_frames=2, max_text_len=12, max_beams=4, stop="never")
    x = make_frames(4, 2, S, 31)
    cam = _camera(3, 2, 260, 300, 5)
    from gitcap.preprocess import preprocess_frames
    cam_pre = preprocess_frames(cam, S)
    kw = dict(beam_size=4, max_steps=10)
    want_x = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in m.infer(x.cuda(), on_device=True, **kw).items()}
    want_c = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in m.infer(cam_pre, on_device=True, **kw).items()}
    futs = []
    for i in range(9):
        src = (x, cam, x.clone().pin_memory(), cam.cuda())[i % 4]
        futs.append((i % 4, m.infer_async(src, save_logits=(i == 4), visual_features=(i == 4), **kw)))
    for k, f in futs:
        r = f.result()
        w = want_x if k in (0, 2) else want_c
        assert torch.equal(r["predictions"].cpu(), w["predictions"].cpu()), k
        assert torch.equal(r["logprobs"].cpu(), w["logprobs"].cpu()), k
    r4 = futs[4][1].result()
    _, vis = m.forward_image_enc(x.cuda())
    assert torch.equal(r4["visual_features"].cpu(), vis.cpu())
    # the synchronous search on raw frames (no gitcap_beam_search_raw: submit + wait) and on a CPU tensor
    assert torch.equal(m.infer(cam, **kw)["predictions"].cpu(), want_c["predictions"].cpu())
    assert torch.equal(m.infer(x, **kw)["predictions"].cpu(), want_x["predictions"].cpu())
    # teacher forward from a CPU tensor of 11 clips (chunks of max_batch through the ring) == from the device
    big = torch.cat([x, x.flip(0), x[:3]], 0)
    a, b = m.teacher_forward(big, **kw), m.teacher_forward(big.cuda(), **kw)
    assert len(a) == len(b) == 11
    for p, q in zip(a, b):
        assert torch.equal(p["predictions"].cpu(), q["predictions"].cpu()) and torch.equal(p["output"].cpu(), q["output"].cpu())
        assert torch.equal(p["visual_features"].cpu(), q["visual_features"].cpu())


def test_host_fed_bench_shape_bitwise(captioner_cls):
    """The bench's host-fed case at its real shape (16 x 6-frame clips, GIT-base, uint8 224 x 224 page-locked frames, 3 in
    flight, more batches than ring entries): ids bitwise the synchronous raw call's."""
    cfg = git_base(6)
    m = captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=16, max_frames=6, max_text_len=20, stop="never")
    ins = [_camera(16, 6, 224, 224, 40 + i).pin_memory() for i in range(3)]
    want = [m.greedy_decode(x.cuda(), max_len=20).cpu().clone() for x in ins]
    pend, bad = [], []
    for i in range(10):
        pend.append((i % 3, m.greedy_decode_async(ins[i % 3], max_len=20)))
        if len(pend) == 3:
            k, f = pend.pop(0)
            if not torch.equal(f.result(), want[k]):
                bad.append(i)
    for k, f in pend:
        if not torch.equal(f.result(), want[k]):
            bad.append(("tail", k))
    assert not bad, bad


def test_sync_call_between_submit_and_result_with_exchange_failure(captioner_cls):
    """a submission that a synchronous call's drain has already stream-waited -- but whose rows no future has
    handed out yet -- must still be marked when the failed statistics exchange is reported by that synchronous call; its future
    re-runs the batch instead of returning undefined ids."""
    from gitcap import _lib
    lib = _lib.load()
    cfg = git_base(6)
    w = synthetic_weights(cfg, 0)
    frs = [make_frames(10, 6, cfg.image_size, 23 + i) for i in range(2)]          # 11 820 image rows: fused epilogues on 256-row tiles
    m = captioner_cls(cfg, w, max_batch=10, max_frames=6, max_text_len=8, stop="never")
    want = [m.greedy_decode(f.cuda(), max_len=8).cpu() for f in frs]
    m.poll_errors()
    old = lib.gitcap_dbg_config(6, 1)
    try:
        fut = m.greedy_decode_async(frs[0].cuda(), max_len=8)      # device in -> device out: result() alone cannot vouch
        got_sync = m.greedy_decode(frs[1], max_len=8)              # CPU in -> CPU out: drains (stream-waits `fut`), then vouches
        assert torch.equal(got_sync, want[1])
        assert fut._sub.poisoned, "the waited-but-undelivered submission was not marked"
        assert torch.equal(fut.result().cpu(), want[0])
        m.poll_errors()
        assert not m._undelivered
    finally:
        lib.gitcap_dbg_config(6, old)


def test_refused_submissions_leave_the_pipeline_usable(captioner_cls):
    """an entry point that refuses a submission -- bad frame sizes, a batch beyond max_batch -- between good ones: the
    refusal is reported, no ticket is consumed, the submissions around it deliver the synchronous results, and the slot sequence goes on."""
    import ctypes
    from gitcap import _lib
    lib = _lib.load()
    cfg = git_tiny(2)
    m = captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=2, max_frames=2, max_text_len=10, stop="never")
    x = make_frames(2, 2, cfg.image_size, 11).cuda()
    cam = _camera(2, 2, 80, 96, 4).cuda()
    want_x, want_c = m.greedy_decode(x, max_len=9).clone(), m.greedy_decode(cam, max_len=9).clone()
    ids = torch.empty((4, 10), dtype=torch.int64, device="cuda")
    steps = torch.zeros((1,), dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    tk = ctypes.c_int(-7)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    futs = [m.greedy_decode_async(x, max_len=9), m.greedy_decode_async(cam, max_len=9)]
    assert lib.gitcap_greedy_raw_submit(m._handle, p(cam), 2, 2, 0, 96, 9, 0, p(ids), p(steps), st, ctypes.byref(tk)) == -1      # H = 0
    assert b"raw frames" in lib.gitcap_last_error(m._handle) and tk.value == -7
    assert lib.gitcap_greedy_raw_submit(m._handle, p(cam), 4, 2, 80, 96, 9, 0, p(ids), p(steps), st, ctypes.byref(tk)) == -1     # B > max_batch
    assert lib.gitcap_greedy_submit(m._handle, p(x), 2, 3, 9, 0, p(ids), p(steps), st, ctypes.byref(tk)) == -1                    # F > max_frames
    assert lib.gitcap_greedy_raw_submit(m._handle, None, 2, 2, 80, 96, 9, 0, p(ids), p(steps), st, ctypes.byref(tk)) == -1
    assert tk.value == -7
    futs += [m.greedy_decode_async(cam.cpu(), max_len=9), m.greedy_decode_async(x.cpu(), max_len=9), m.greedy_decode_async(x, max_len=9)]
    got = [f.result() for f in futs]
    assert torch.equal(got[0], want_x) and torch.equal(got[1], want_c) and torch.equal(got[2], want_c.cpu())
    assert torch.equal(got[3], want_x.cpu()) and torch.equal(got[4], want_x)
    assert torch.equal(m.greedy_decode(x, max_len=9), want_x)
    m.poll_errors()
