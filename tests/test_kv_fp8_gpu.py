"""kv_cache="v_e4m3" (include/gitcap.h: gitcap_set_kv_cache; north_star: "a KV cache for the decode loop ... bf16/fp8"): the token loop
reads the V rows of the image prefix as e4m3 codes with one power-of-two scale per (token, head).  The reference has no cache at all
(src/models/model.py:412-418 recomputes the prefix per token; the step is bound at :442-445, :519), so the contract is the oracle with
the same rounding point, GitOracle(emulate_fp8_v=True), on both weight families:

  * every decoder layer's TEXT rows against that oracle on the device's own layer input, under the FIXED single-stage tolerances of
    tests/test_stress_layers_gpu.py (ulps of the row maximum), and at most half as far from it as from the bf16-V oracle (what the
    mode costs in that layer): the mode's oracle pins the kernel;
  * end to end on plain weights within 3 x the bf16 tolerance of its own oracle and within the 0.3 bar of the bf16-compute oracle;
  * the bitwise properties of the default mode: cached step == teacher-forced pass, batch invariance, 8- and 16-wave forms of the
    text attention, pipelined == synchronous, device beam search == host operator."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_layers import device_stages, ulps_of_rowmax                                   # noqa: E402
from test_stress_layers_gpu import TOL_DEC_MAX_ULP, TOL_RMS_ULP                            # noqa: E402

from gitcap.config import git_base                                                         # noqa: E402
from gitcap.weights import stress_weights, synthetic_weights                               # noqa: E402
from oracle.git_oracle import GitOracle, make_frames                                       # noqa: E402

pytestmark = pytest.mark.gpu
LOGIT_TOL_EMUL = 0.08                    # tests/test_parity_gpu.py


@pytest.fixture(scope="module")
def captioner_cls():
    from gitcap.model import GitCaptioner
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return GitCaptioner


def _rms(t):
    return float(t.double().pow(2).mean().sqrt())


@pytest.mark.parametrize("family", ["plain", "stress"])
def test_v_e4m3_text_rows_track_their_oracle_per_layer(captioner_cls, family):
    cfg = git_base(2)
    w = (stress_weights if family == "stress" else synthetic_weights)(cfg, 0)
    fr = make_frames(2, 2, cfg.image_size, 1234)
    g = torch.Generator().manual_seed(20)
    ids = torch.randint(1000, cfg.vocab_size, (2, 20), generator=g)
    ids[:, 0] = cfg.cls_token_id
    m = captioner_cls(cfg, w, max_batch=2, max_frames=2, max_text_len=24, kv_cache="v_e4m3")
    dev = device_stages(m, fr, ids)
    own = GitOracle(cfg, w, emulate_bf16=True, emulate_fp8_v=True)
    bf = GitOracle(cfg, w, emulate_bf16=True)
    S_img = 2 * cfg.tokens_per_frame
    hid = dev["hidden"]
    rows = []
    with torch.no_grad():
        for l in range(cfg.dec_layers):
            got = hid[:, l + 1, S_img:]
            a, b = own.dec_layer_text(l, hid[:, l], S_img), bf.dec_layer_text(l, hid[:, l], S_img)
            rows.append((l, ulps_of_rowmax(got, a), _rms(got - a), _rms(got - b)))
        # the image rows are untouched by the mode: same single-stage tolerance against the plain emulating oracle
        for l in range(cfg.dec_layers):
            e = ulps_of_rowmax(hid[:, l + 1, :S_img], bf.dec_layer_img(l, hid[:, l, :S_img]))
            assert e[0] <= TOL_DEC_MAX_ULP and e[1] <= TOL_RMS_ULP, (family, "image rows", l, e)
        l_own = own.forward_output_logits(fr, ids)[0]
        l_bf = bf.forward_output_logits(fr, ids)[0]
    for l, e, r_own, r_mode in rows:
        print(f"v_e4m3 {family} dec layer {l} text rows: {e[0]:.3f} ulp of row max (rms {e[1]:.4f}); rms device - own oracle {r_own:.5f}, "
              f"device - bf16-V oracle {r_mode:.5f}")
        assert e[0] <= TOL_DEC_MAX_ULP and e[1] <= TOL_RMS_ULP, (family, l, e)
        assert r_own < 0.5 * r_mode, (family, l, r_own, r_mode)
    lg = dev["logits"]
    d_own, d_bf = float((lg - l_own).abs().max()), float((lg - l_bf).abs().max())
    print(f"v_e4m3 {family}: logits max |device - own oracle| {d_own:.3f}, |device - bf16-V oracle| {d_bf:.3f} (= what the mode costs), "
          f"|own - bf16-V oracle| {float((l_own - l_bf).abs().max()):.3f}, logit std {float(l_bf.std()):.2f}")
    if family == "plain":
        assert d_own < 3 * LOGIT_TOL_EMUL
        assert d_bf < 0.3
    # the default handle is untouched by the mode's existence: a bf16 handle gives the bf16 bits
    m0 = captioner_cls(cfg, w, max_batch=2, max_frames=2, max_text_len=24)
    assert not torch.equal(m0(fr, ids).cpu(), lg)


def test_v_e4m3_keeps_the_bitwise_properties(captioner_cls):
    from gitcap import _lib
    lib = _lib.load()
    cfg = git_base(6)
    w = stress_weights(cfg, 0)
    m = captioner_cls(cfg, w, max_batch=5, max_frames=6, max_text_len=16, max_beams=4, kv_cache="v_e4m3", stop="never")
    fr = make_frames(5, 6, cfg.image_size, 61).cuda()
    out5 = m.greedy_decode(fr, max_len=14)
    out2 = m.greedy_decode(fr[:2], max_len=14)
    out1 = m.greedy_decode(fr[3:4], max_len=14)
    assert torch.equal(out2, out5[:2]) and torch.equal(out1[0], out5[3])                # batch invariance (row prologue / row kernels)
    _, vis = m.forward_image_enc(fr)
    tf = m.forward_decoder(out5[:, :-1], vis)                                           # teacher-forced: T = 14 positions at once
    assert torch.equal(tf.argmax(-1), out5[:, 1:])
    for t in range(14):
        assert torch.equal(m.step_logits(out5[:, t], t), tf[:, t]), t                   # cached step == teacher-forced pass, bitwise
    # pipelined == synchronous
    futs = [m.greedy_decode_async(fr if i % 2 == 0 else fr[:2], max_len=14) for i in range(5)]
    for i, f in enumerate(futs):
        assert torch.equal(f.result(), out5 if i % 2 == 0 else out2), i
    # device-resident beam search == host operator over the same kernels (the beams of a clip share its image K / V codes)
    dev = m.infer(fr[:4], beam_size=4, max_steps=12, on_device=True)
    host = m.infer(fr[:4], beam_size=4, max_steps=12, on_device=False)
    assert torch.equal(dev["predictions"], host["predictions"])
    # more (row, head) units than CUs: the 8-wave form of the text attention == the 16-wave form (speed switch 9)
    cfg1 = git_base(0)
    m1 = captioner_cls(cfg1, stress_weights(cfg1, 0), max_batch=24, max_frames=1, max_text_len=12, kv_cache="v_e4m3", stop="never")
    fr1 = make_frames(24, 1, cfg1.image_size, 29).cuda()
    a = m1.greedy_decode(fr1, max_len=12).clone()
    old = lib.gitcap_dbg_config(9, 0)
    try:
        b = m1.greedy_decode(fr1, max_len=12).clone()
    finally:
        lib.gitcap_dbg_config(9, old)
    assert torch.equal(a, b)
    _, v1 = m1.forward_image_enc(fr1)
    assert torch.equal(m1.forward_decoder(a[:, :-1], v1).argmax(-1), a[:, 1:])


def test_v_e4m3_argument_checks(captioner_cls):
    cfg = git_base(2)
    with pytest.raises(ValueError, match="kv_cache"):
        captioner_cls(cfg, synthetic_weights(cfg, 0), max_batch=1, kv_cache="k_e4m3")
