"""Pipelined submissions through the C ABI (gitcap_greedy_submit / gitcap_beam_search_submit / _wait): the overlapped form of the
device-resident beam search (BASELINE configs[4]; the reference's teacher call, src/models/model.py:762-768 with the defaults of
:702-708 and the search of :479-678) and what happens to submissions in flight when the LayerNorm statistics exchange fails
soft (include/gitcap.h: gitcap_poll_errors)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from gitcap.config import git_base, git_large, git_tiny
from gitcap.weights import quantize_weights_fp8, synthetic_weights
from oracle.git_oracle import make_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def captioner_cls():
    from gitcap.model import GitCaptioner
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return GitCaptioner


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_pipelined_beam_search_bitwise(captioner_cls, size):
    """Searches of several batches in flight on the library's streams (each slot has its own beam state, text K/V and row
    workspace), mixed with greedy submissions and ragged batches: every result bitwise the synchronous gitcap_beam_search's;
    the exported per-step logits equal the host operator's saved logits; visual features equal gitcap_encode's."""
    cfg = git_tiny(2) if size == "tiny" else git_base(2)
    w = synthetic_weights(cfg, 0)
    m = captioner_cls(cfg, w, max_batch=4, max_frames=2, max_text_len=12, max_beams=4, stop="never")
    inputs = [make_frames(b, 2, cfg.image_size, 700 + i).cuda() for i, b in enumerate((4, 3, 4, 1, 2))]
    want = [m.infer(x, beam_size=4, max_steps=10, on_device=True) for x in inputs]
    want = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in r.items()} for r in want]
    greedy = [m.greedy_decode(x, max_len=8).clone() for x in inputs]
    torch.cuda.synchronize()
    pend, bad = [], []
    for i in range(24):
        k = (i * 3 + i // 5) % len(inputs)
        if i % 4 == 3:
            pend.append(("g", k, m.greedy_decode_async(inputs[k], max_len=8)))
        else:
            pend.append(("b", k, m.infer_async(inputs[k], beam_size=4, max_steps=10)))
        while len(pend) >= 4 - (i % 3 == 0):
            kind, k0, f = pend.pop(0)
            r = f.result()
            ok = torch.equal(r, greedy[k0]) if kind == "g" else (torch.equal(r["predictions"], want[k0]["predictions"]) and
                                                                   torch.equal(r["logprobs"], want[k0]["logprobs"]))
            if not ok:
                bad.append((i, kind, k0))
    for kind, k0, f in pend:
        r = f.result()
        ok = torch.equal(r, greedy[k0]) if kind == "g" else torch.equal(r["predictions"], want[k0]["predictions"])
        if not ok:
            bad.append(("tail", kind, k0))
    assert not bad, bad
    # per-step logits + visual features out of the pipelined call
    f = m.infer_async(inputs[0], beam_size=4, max_steps=10, save_logits=True, visual_features=True)
    g = m.greedy_decode_async(inputs[2], max_len=8)
    r = f.result()
    assert torch.equal(r["predictions"], want[0]["predictions"]) and torch.equal(g.result(), greedy[2])
    host = m.infer(inputs[0], beam_size=4, max_steps=10, on_device=False, save_logits=True)
    assert torch.equal(host["predictions"], r["predictions"])
    steps = r["logits_dict"]
    assert tuple(steps.shape) == (9, 16, cfg.vocab_size)
    for t, lg in enumerate(host["logits_dict"]):                  # the host operator may stop early (model.py:640)
        assert np.array_equal(np.asarray(lg), steps[t].cpu().numpy()), t
    _, vis = m.forward_image_enc(inputs[0])
    assert torch.equal(vis, r["visual_features"])
    # a synchronous call afterwards, and the synchronous search with logits / features through the same entry point
    assert torch.equal(m.infer(inputs[1], beam_size=4, max_steps=10)["predictions"], want[1]["predictions"])
    r2 = m.infer(inputs[0], beam_size=4, max_steps=10, on_device=True, save_logits=True)
    assert torch.equal(r2["logits_dict"], steps)
    # teacher forward: the pipelined device path == the host-operator path, clip by clip, also across chunks of max_batch
    x = torch.cat([inputs[0], inputs[1], inputs[2]], 0)           # 11 clips through max_batch 4
    dev = m.teacher_forward(x, beam_size=4, max_steps=10)
    hst = m.teacher_forward(x, beam_size=4, max_steps=10, on_device=False)
    assert len(dev) == len(hst) == 11
    for a, b in zip(dev, hst):
        assert torch.equal(a["predictions"], b["predictions"]) and torch.equal(a["output"], b["output"])
        assert torch.equal(a["visual_features"], b["visual_features"])
        assert torch.allclose(a["logprobs"].reshape(-1), b["logprobs"].reshape(-1), atol=1e-5)     # (host operator: log-softmax by torch)


def test_config4_exact_fixture_pipelined(captioner_cls, golden_dir):
    """BASELINE configs[4] at its real shape (GIT-large, 10 frames, e4m3 weights, beam 4, 15 steps) through the pipelined path:
    four clips, three submissions in flight, each must reproduce the certified caption of tests/golden/cfg4_beam_exact.npz for
    the fixture's clip and the synchronous search's result for the others -- in bf16 compute and with compute="fp8_ffn"."""
    g = np.load(os.path.join(golden_dir, "cfg4_beam_exact.npz"))
    F, beams, steps = int(g["frames"]), int(g["beams"]), int(g["max_steps"])
    cfg = git_large(num_frames=F)
    wq = quantize_weights_fp8(synthetic_weights(cfg, int(g["weight_seed"])))
    fix = make_frames(1, F, cfg.image_size, int(g["frame_seed"]))
    others = make_frames(3, F, cfg.image_size, 4242)
    batch = torch.cat([others[:2], fix, others[2:]], 0).cuda()       # the fixture's clip is row 2 of a 4-clip batch
    kw = dict(beam_size=beams, max_steps=steps, length_penalty=float(g["length_penalty"]), per_node_beam_size=int(g["per_node_beam_size"]))
    for compute in ("bf16", "fp8_ffn"):
        m = captioner_cls(cfg, wq, max_batch=4, max_frames=F, max_text_len=16, max_beams=beams, weight_dtype="fp8_e4m3", compute=compute)
        want = m.infer(batch, on_device=True, **kw)
        want_p, want_l = want["predictions"].clone(), want["logprobs"].clone()
        if compute == "bf16":
            assert torch.equal(want_p[2].cpu(), torch.from_numpy(g["predictions"])[0])
            assert abs(float(want_l[2, 0]) - float(g["logprob"])) < 0.05
        solo = m.infer(batch[2:3], on_device=True, **kw)            # batch invariance of the search
        assert torch.equal(solo["predictions"][0], want_p[2])
        futs = [m.infer_async(batch if i % 2 == 0 else batch[2:3], **kw) for i in range(7)]
        for i, f in enumerate(futs):
            r = f.result()
            if i % 2 == 0:
                assert torch.equal(r["predictions"], want_p) and torch.equal(r["logprobs"], want_l), (compute, i)
            else:
                assert torch.equal(r["predictions"][0], want_p[2]), (compute, i)
        del m


def test_exchange_failure_poisons_submissions_in_flight(captioner_cls):
    """when a fused GEMM + LayerNorm launch gives up waiting (forced: gitcap_dbg_config(6, 1)), every submission in flight
    holds undefined ids.  Their tickets are marked in the library -- gitcap_greedy_wait returns GITCAP_ERR_EXCHANGE for them every
    time, a retry cannot hand the ids out -- and the Python futures re-run their batches on the handle (which has switched to the
    unfused launches): the caller sees correct results, whichever future it asks first, however often."""
    from gitcap import _lib
    lib = _lib.load()
    cfg = git_base(6)
    w = synthetic_weights(cfg, 0)
    frs = [make_frames(10, 6, cfg.image_size, 23 + i) for i in range(3)]          # 11 820 image rows: fused epilogues on 256-row tiles
    m = captioner_cls(cfg, w, max_batch=10, max_frames=6, max_text_len=8, max_beams=2, stop="never")
    want = [m.greedy_decode(f.cuda(), max_len=8).cpu() for f in frs]
    want_b = m.infer(frs[0].cuda(), beam_size=2, max_steps=6)["predictions"].cpu()
    m.poll_errors()
    old = lib.gitcap_dbg_config(6, 1)
    assert old == 0
    try:
        futs = [m.greedy_decode_async(f, max_len=8) for f in frs]                 # CPU in -> CPU out: result() can vouch
        fb = m.infer_async(frs[0], beam_size=2, max_steps=6)
        tickets = [f._sub.ticket for f in futs]
        got1 = futs[1].result()                                                    # asked out of order
        assert torch.equal(got1, want[1])
        assert torch.equal(futs[1].result(), want[1])                              # asking again gives the same tensor
        # the other submissions were in flight at the failure: the library refuses their tickets, every time
        for _ in range(2):
            rc = lib.gitcap_greedy_wait(m._handle, tickets[0], None)
            assert rc == _lib.ERR_EXCHANGE, rc
        assert torch.equal(futs[0].result(), want[0]) and torch.equal(futs[2].result(), want[2])
        assert torch.equal(fb.result()["predictions"].cpu(), want_b)
        m.poll_errors()                                                            # clean: reported once, handle degraded
        # the handle carries on (unfused launches, same bits), pipelined and synchronous
        futs = [m.greedy_decode_async(f, max_len=8) for f in frs]
        for f, x in zip(futs, want):
            assert torch.equal(f.result(), x)
        assert torch.equal(m.greedy_decode(frs[2], max_len=8), want[2])
    finally:
        lib.gitcap_dbg_config(6, old)
    # device tensors in, no host synchronisation inside result(): nothing can be vouched for there -- the documented contract is
    # poll_errors() after the caller's own synchronisation; a failure then poisons what is still in flight
    m2 = captioner_cls(cfg, w, max_batch=10, max_frames=6, max_text_len=8, stop="never")
    old = lib.gitcap_dbg_config(6, 1)
    try:
        f0 = m2.greedy_decode_async(frs[0].cuda(), max_len=8)
        f1 = m2.greedy_decode_async(frs[1].cuda(), max_len=8)
        r0 = f0.result()                                                           # device tensor: no synchronisation inside
        torch.cuda.synchronize()
        try:
            m2.poll_errors()
            late = False                  # the failure was already seen when f1 was submitted: f0 was marked then and re-run
        except _lib.GitcapExchangeTimeout:
            late = True                   # seen only now: r0 is undefined (the caller was told), f1 is marked
        if not late:
            assert torch.equal(r0.cpu(), want[0])
        assert torch.equal(f1.result().cpu(), want[1])
        assert torch.equal(m2.greedy_decode(frs[0].cuda(), max_len=8).cpu(), want[0])
    finally:
        lib.gitcap_dbg_config(6, old)
