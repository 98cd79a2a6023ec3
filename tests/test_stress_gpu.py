"""Round-5 parity hardening, through the C ABI on the MI355X box:

  (a) a second weight family with the statistics of trained checkpoints (gitcap.weights.stress_weights: outlier LayerNorm
      channels, saturating GELU inputs, large-norm CLS / position rows, a peaked head) against the bf16-emulating oracle and
      the HF-transformers fixtures tests/golden/hf_*_stress.npz (oracle/gen_golden_hf.py stress);
  (b) the caption lengths the reference's callers really use: teacher-forced T = 39 (src/utils/tokenizer.py:5-27: CLS + up
      to 38 ids) and greedy max_len = y.shape[-1] + 5 = 44 (src/inference.py:51), the student decoder at 44 and at its
      limit 63;
  (c) compute="fp8_ffn" can saturate but not silently: clamped codes are counted (gitcap_fp8_saturations) and the scale is
      settable (gitcap_set_fp8_scale) -- on the stress weights the mode either passes at a calibrated scale or reports
      saturations, never neither.

Tolerances (end to end; the stage-by-stage checks on the device's own inputs, with FIXED tolerances, are
tests/test_stress_layers_gpu.py -- they are what localises a kernel error; the numbers here bound what the network makes of it).
The plain-weight rules of tests/test_parity_gpu.py are 0.08 against the bf16-emulating oracle and 0.20 against fp32 on logits of
std ~4, where the emulating oracle itself sits 0.10 from fp32.  Outlier channels amplify that distance (a bf16 rounding step of a
value of 60 is 0.25) and two legitimate roundings of an ill-conditioned row decorrelate, so here (`_tols`, `_check_logits`):
  * max rule: |device - emulating oracle| <= 1.5 x and |device - fp32| <= 2 x the distance d = max |emulating - fp32| measured on
    the input at hand, never below the plain-weight numbers;
  * rms rule: the device no further (rms) from either oracle than 1.5 x the emulating oracle's own rms distance from fp32;
  * PINS (round 6): on the fixed fixtures the max and rms device-vs-emulating distances are additionally held to their
    measured values + headroom (`PINS`), so that a regression of the size of the bf16 emulation error itself cannot hide behind a
    tolerance derived from the same input; the near-tie gate of the token checks is capped at an absolute 2.0 logits.
"""
import os

import numpy as np
import pytest
import torch

from gitcap.config import GitCapConfig, git_base, git_tiny
from gitcap.weights import quantize_weights_fp8, stress_weights, synthetic_weights
from oracle.git_oracle import GitOracle, make_frames

pytestmark = pytest.mark.gpu

LOGIT_TOL_EMUL, LOGIT_TOL_FP32, NEAR_TIE = 0.08, 0.20, 0.16      # tests/test_parity_gpu.py


@pytest.fixture(scope="module")
def captioner_cls():
    from gitcap.model import GitCaptioner
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return GitCaptioner


def _tols(l_e, l_f):
    d = float((l_e - l_f).abs().max())
    return max(LOGIT_TOL_EMUL, 1.5 * d), max(LOGIT_TOL_FP32, 2.0 * d), d


def _rms(x):
    return float(x.double().pow(2).mean().sqrt())


# measured (GPUTEST r05 / r06 logs: max, rms of device - emulating oracle) -> allowed (about 1.4 x)
PINS = {"tiny stress": (0.45, 0.050), "base stress": (1.50, 0.135), "stress T=39": (1.65, 0.130), "plain T=39": (0.15, 0.027),
        "GIT-large stress": (1.50, 0.135)}
NEAR_TIE_CAP = 2.0


def _check_logits(tag, lg, l_e, l_f):
    """max rule: |device - emulating oracle| <= 1.5 x and |device - fp32| <= 2 x the distance the bf16 rounding points alone create
    on this input (never below the plain-weight tolerances); rms rule: over all logits the device is no further from the
    emulating oracle (and from fp32) than two independent roundings of that size would be (sqrt(2) x the oracle's own rms distance
    from fp32, taken as 1.5 x; measured 0.7 - 1.2 x) -- a kernel that mishandles large magnitudes moves the rms, two legitimate
    roundings of an ill-conditioned row move only the max."""
    tol_e, tol_f, d = _tols(l_e, l_f)
    de, df = float((lg - l_e).abs().max()), float((lg - l_f).abs().max())
    re, rf, r0 = _rms(lg - l_e), _rms(lg - l_f), _rms(l_e - l_f)
    print(f"{tag}: max |emul - fp32| {d:.3f}, device vs emul {de:.3f} (tol {tol_e:.3f}), vs fp32 {df:.3f} (tol {tol_f:.3f}); "
          f"rms emul - fp32 {r0:.4f}, device - emul {re:.4f}, device - fp32 {rf:.4f}; logit std {float(l_f.std()):.2f}")
    assert de < tol_e and df < tol_f
    assert re < 1.5 * r0 + 0.004 and rf < 1.5 * r0 + 0.004
    if tag in PINS:
        assert de < PINS[tag][0] and re < PINS[tag][1], (tag, de, re, PINS[tag])
    return min(tol_e, NEAR_TIE_CAP / 2), tol_f


def _margin_gated(dev_ids, logits, near_tie):
    """`logits` = the oracle teacher-forced on the device's own tokens: every device token is the oracle's argmax or lies
    within near_tie of it.  Returns the fraction of exact agreements."""
    chosen = logits.gather(2, dev_ids[:, 1:, None]).squeeze(-1)
    gap = logits.max(-1).values - chosen
    assert float(gap.max()) < near_tie, f"device token outside a near-tie of the oracle: gap {gap.max():.3f} (allowed {near_tie:.3f})"
    return float((gap == 0).float().mean())


def test_tiny_stress_stages_vs_oracle_and_hf(captioner_cls, golden_dir):
    cfg = git_tiny(2)
    w = stress_weights(cfg, 0)
    g = np.load(os.path.join(golden_dir, "hf_tiny_stress.npz"))
    fr = make_frames(2, 2, cfg.image_size, int(g["frame_seed"]))
    m = captioner_cls(cfg, w, max_batch=4, max_text_len=16)
    emul, fp32 = GitOracle(cfg, w, emulate_bf16=True), GitOracle(cfg, w)
    _, vis = m.forward_image_enc(fr)
    vis = vis.cpu()
    v_e, v_f = emul.encode_frames(fr), fp32.encode_frames(fr)
    dv = float((v_e - v_f).abs().max())
    assert float(vis.abs().max()) > 8.0                                        # the outliers reach the device
    assert (vis - v_e).abs().max() < max(2e-2, 1.5 * dv), (float((vis - v_e).abs().max()), dv)
    assert (vis - v_f).abs().max() < 2.0 * dv + 1e-2
    assert _rms(vis - v_e) < 1.5 * _rms(v_e - v_f) + 1e-3
    assert np.abs(vis.numpy() - g["visual"]).max() < 2.0 * dv + 1e-2          # HF fp32 fixture
    ids = torch.from_numpy(g["prefix_ids"])
    lg = m(fr, ids).cpu()
    l_e, _ = emul.forward_output_logits(fr, ids)
    l_f, _ = fp32.forward_output_logits(fr, ids)
    tol_e, tol_f = _check_logits("tiny stress", lg, l_e, l_f)
    assert np.abs(lg.numpy() - g["logits"]).max() < tol_f                      # HF fp32 fixture
    # per-layer hidden states (outlier channels of ~60 in the residual stream) against the oracle and the HF fixture
    _, _, hid = m.forward_output_logits(fr, ids, output_hidden_states=True)
    got = torch.stack([h.cpu() for h in hid], 1)
    _, mem = emul.forward_image_enc(fr)
    _, want = emul.decoder_full(mem, ids, return_hidden=True)
    want = torch.stack(want, 0)
    _, mem_f = fp32.forward_image_enc(fr)
    _, want_f = fp32.decoder_full(mem_f, ids, return_hidden=True)
    dh = float((want - torch.stack(want_f, 0)).abs().max())
    assert float(got.abs().max()) > 20.0
    assert (got - want).abs().max() < max(0.06, 1.5 * dh), (float((got - want).abs().max()), dh)
    assert np.abs(got.numpy() - g["hidden"]).max() < max(0.12, 2.0 * dh)
    # tokens: margin-gated against the emulating oracle, and against the HF golden ids until the first near-tie
    out = m.greedy_decode(fr, max_len=8, stop="never").cpu()
    le, _ = emul.forward_output_logits(fr, out[:, :-1])
    _margin_gated(out, le, 2 * tol_e)
    gold = torch.from_numpy(g["greedy_ids"])
    margin = torch.from_numpy(g["greedy_top_vals"][..., 0] - g["greedy_top_vals"][..., 1])
    for b in range(2):
        for t in range(8):
            if out[b, t + 1] != gold[b, t + 1]:
                assert margin[b, t] < tol_f, (b, t, float(margin[b, t]))
                break
    # the exact-KV-cache and batch-invariance properties hold on these weights too (bitwise)
    _, v2 = m.forward_image_enc(fr)
    tf = m.forward_decoder(out[:, :-1].cuda(), v2)
    assert torch.equal(tf.argmax(-1).cpu(), out[:, 1:])
    assert torch.equal(m.greedy_decode(fr[1:2], max_len=8, stop="never").cpu()[0], out[1])


def test_base_stress_vs_hf_golden(captioner_cls, golden_dir):
    """GIT-base, 2 clips x 2 frames, 20 greedy tokens on the stress weights against the HF fp32 run and the emulating oracle."""
    cfg = git_base(2)
    w = stress_weights(cfg, 0)
    g = np.load(os.path.join(golden_dir, "hf_base_F2_stress.npz"))
    fr = make_frames(2, 2, cfg.image_size, int(g["frame_seed"]))
    m = captioner_cls(cfg, w, max_batch=2, max_frames=2, max_text_len=24)
    gold = torch.from_numpy(g["greedy_ids"])
    top_i, top_v = torch.from_numpy(g["greedy_top_ids"]), torch.from_numpy(g["greedy_top_vals"])
    emul, fp32 = GitOracle(cfg, w, emulate_bf16=True), GitOracle(cfg, w)
    with torch.no_grad():
        l_e, v_e = emul.forward_output_logits(fr, gold[:, :-1])
        l_f, v_f = fp32.forward_output_logits(fr, gold[:, :-1])
    dv = float((v_e - v_f).abs().max())
    _, vis = m.forward_image_enc(fr)
    assert float(vis.abs().max()) > 10.0
    assert (vis.cpu() - v_e).abs().max() < max(2e-2, 1.5 * dv), (float((vis.cpu() - v_e).abs().max()), dv)
    assert _rms(vis.cpu() - v_e) < 1.5 * _rms(v_e - v_f) + 1e-3
    assert np.abs(vis.cpu()[:, ::97, :32].numpy() - g["visual_slice"]).max() < 2.0 * dv + 1e-2
    lg = m.forward_decoder(gold[:, :-1], vis).cpu()
    tol_e, tol_f = _check_logits("base stress", lg, l_e, l_f)
    dg = (torch.gather(lg, 2, top_i) - top_v).abs()                            # HF fp32 fixture: the top-8 logits of each step
    assert dg.max() < tol_f and dg.mean() < 0.25 * tol_f
    margin = top_v[..., 0] - top_v[..., 1]
    agree = lg.argmax(-1) == gold[:, 1:]
    assert bool(agree[margin > tol_f].all())
    out = m.greedy_decode(fr, max_len=20, stop="never").cpu()
    for b in range(2):
        for t in range(20):
            if out[b, t + 1] != gold[b, t + 1]:
                assert margin[b, t] < tol_f, (b, t, float(margin[b, t]))
                break
    with torch.no_grad():
        le, _ = emul.forward_output_logits(fr, out[:, :-1])
    _margin_gated(out, le, 2 * tol_e)
    assert torch.equal(m.forward_decoder(out[:, :-1].cuda(), vis).argmax(-1).cpu(), out[:, 1:])     # exact KV cache, bitwise


def test_stress_speed_switches_and_pipeline_bitwise(captioner_cls):
    """The equalities the speed switches and the pipeline rest on are arithmetic identities, not properties of benign data:
    on the stress weights (GIT-base, 8 clips x 6 frames) the fused / unfused LayerNorm epilogues, the tile kernels, the fused
    text FFN, the shared-row head and three submissions in flight all give the bits of the default synchronous call."""
    from gitcap import _lib
    lib = _lib.load()
    cfg = git_base(6)
    w = stress_weights(cfg, 0)
    m = captioner_cls(cfg, w, max_batch=8, max_frames=6, max_text_len=12, stop="never")
    fr = make_frames(8, 6, cfg.image_size, 23).cuda()

    def run(n):
        _, vis = m.forward_image_enc(fr[:n])
        ids = m.greedy_decode(fr[:n], max_len=12)
        return vis.clone(), ids.clone(), m.forward_decoder(ids[:, :-1], vis).clone()
    base8, base1 = run(8), run(1)
    assert torch.equal(base1[1], base8[1][:1]) and torch.equal(base1[0], base8[0][:1])
    assert torch.isfinite(base8[2]).all() and len(set(base8[1][0].tolist())) > 4
    for key, value in [(0, 0), (2, 1 << 30), (3, 0), (2, 1), (5, 0), (7, 0), (10, 0), (11, 0)]:
        old = lib.gitcap_dbg_config(key, value)
        try:
            for base, n in ((base8, 8), (base1, 1)):
                for a, b in zip(base, run(n)):
                    assert torch.equal(a, b), (key, value, n)
        finally:
            lib.gitcap_dbg_config(key, old)
    futs = [m.greedy_decode_async(fr, max_len=12) for _ in range(3)]
    for f in futs:
        assert torch.equal(f.result(), base8[1])


@pytest.mark.parametrize("family", ["plain", "stress"])
def test_caller_lengths_teacher_forced_39_and_greedy_44(captioner_cls, family):
    """The reference's callers: forward_output_logits(x, y) with y of up to 39 ids (src/utils/tokenizer.py:5-27) and
    greedy_decode(x, max_len=y.shape[-1] + 5) = 44 steps (src/inference.py:51), GIT-base, 6-frame clips.  Teacher-forced
    logits at T = 39 against the oracle; 44 greedy steps margin-gated; the KV-cached loop == one teacher-forced pass
    BITWISE at every one of the 44 steps (2 rows: the row-prologue form; 5 rows: the row kernels)."""
    cfg = git_base(6)
    w = (stress_weights if family == "stress" else synthetic_weights)(cfg, 0)
    m = captioner_cls(cfg, w, max_batch=5, max_frames=6, max_text_len=45)
    fr = make_frames(5, 6, cfg.image_size, 61)
    emul, fp32 = GitOracle(cfg, w, emulate_bf16=True), GitOracle(cfg, w)
    g = torch.Generator().manual_seed(39)
    y = torch.randint(1000, cfg.vocab_size, (2, 39), generator=g)
    y[:, 0] = cfg.cls_token_id
    _, vis = m.forward_image_enc(fr[:2])
    lg = m.forward_decoder(y, vis).cpu()
    assert lg.shape == (2, 39, cfg.vocab_size)
    with torch.no_grad():
        _, mem = emul.forward_image_enc(fr[:2])
        ikv = emul.image_kv(mem)
        l_e = emul.decoder_text(ikv, y)
        _, mem_f = fp32.forward_image_enc(fr[:2])
        l_f = fp32.decoder_text(fp32.image_kv(mem_f), y)
    tol_e, tol_f = _check_logits(f"{family} T=39", lg, l_e, l_f)
    # 44 greedy steps: CLS + 44 tokens
    out5 = m.greedy_decode(fr.cuda(), max_len=44, stop="never")
    out = m.greedy_decode(fr[:2].cuda(), max_len=44, stop="never")
    assert out.shape == (2, 45) and torch.equal(out, out5[:2])                 # one/two-row form == row kernels
    with torch.no_grad():
        le = emul.decoder_text(ikv, out[:, :-1].cpu())
    frac = _margin_gated(out.cpu(), le, 2 * tol_e)
    assert frac > 0.8, frac
    # cached == teacher-forced, bitwise, at every step to 44
    _, vis5 = m.forward_image_enc(fr.cuda())
    tf5 = m.forward_decoder(out5[:, :-1], vis5)                                # [5, 44, V]
    assert torch.equal(tf5.argmax(-1), out5[:, 1:])
    for t in range(44):
        assert torch.equal(m.step_logits(out5[:, t], t), tf5[:, t]), t
    _, vis2 = m.forward_image_enc(fr[:2].cuda())
    tf2 = m.forward_decoder(out[:, :-1], vis2)
    assert torch.equal(tf2, tf5[:2])
    for t in range(44):
        assert torch.equal(m.step_logits(out[:, t], t), tf2[:, t]), t
    # beam search at the callers' length: device-resident == host operator over the same kernels
    mb = captioner_cls(cfg, w, max_batch=2, max_frames=6, max_text_len=45, max_beams=4)
    dev = mb.infer(fr[:2], beam_size=4, max_steps=44, on_device=True)
    host = mb.infer(fr[:2], beam_size=4, max_steps=44, on_device=False)
    assert dev["predictions"].shape == (2, 44) and torch.equal(dev["predictions"], host["predictions"])
    assert torch.allclose(dev["logprobs"].cpu(), host["logprobs"].cpu(), atol=1e-5)


@pytest.mark.parametrize("L", [44, 63])
def test_student_decoder_at_caller_lengths(L):
    """The student decoder (model.py:156-187) at greedy max_len 44 (src/inference.py:51) and at its limit 63 (include/gitcap.h:
    max_text_len <= 63): exact KV cache (the cached loop == one teacher-forced pass over its own output) and the oracle."""
    from gitcap.student import StudentCaptioner
    from gitcap.student_config import student_base, student_synthetic_weights
    from oracle.student_oracle import StudentOracle, make_memory
    cfg = student_base()
    w = student_synthetic_weights(cfg, 0)
    m = StudentCaptioner(cfg=cfg, weights=w, device="cuda:0", max_batch=9, max_text_len=L)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, 44)
    ids = m.greedy_decode(mem, max_len=L, stop="never")
    assert ids.shape == (3, L + 1)
    full = m.forward_decoder(ids[:, :-1], mem).cpu()
    assert torch.equal(full.argmax(-1), ids[:, 1:])
    emu = StudentOracle(cfg, w, emulate_bf16=True).forward_decoder(ids[:, :-1], mem)
    assert (full - emu).abs().max().item() < 0.02 * max(1.0, emu.std().item()) * 4      # TOL_EMU of tests/test_student.py (0.08 at std 4)
    top2 = emu.topk(2, dim=-1).values
    sure = (top2[..., 0] - top2[..., 1]) > NEAR_TIE
    assert torch.equal(emu.argmax(-1)[sure], ids[:, 1:][sure])
    assert torch.equal(m.greedy_decode(mem[1:2], max_len=L, stop="never")[0], ids[1])    # batch invariance
    # the device-resident beam search at that length == the host-driven one over full-prefix passes (exact cache: bitwise)
    assert torch.equal(m.beam_search(mem, max_len=L + 1, k=3), m.beam_search_host(mem, max_len=L + 1, k=3))


def _mid768():
    return GitCapConfig(image_size=64, patch_size=16, enc_width=768, enc_layers=2, enc_heads=12, enc_ffn=3072, dec_width=768,
                        dec_layers=2, dec_heads=12, dec_ffn=3072, vocab_size=997, max_text_pos=64, num_frames=3)


def test_fp8_ffn_saturation_is_counted_and_scale_is_settable(captioner_cls):
    """compute="fp8_ffn" quantises the FFN activations of the image rows with ONE static scale (default 1/16: codes cover
    +-28).  On weights with outlier channels and saturating GELU inputs that clamps -- and the clamps are counted on the
    device (gitcap_fp8_saturations), as many as the oracle emulating the mode counts.  With the scale raised until
    nothing clamps (gitcap_set_fp8_scale) the mode is back within a few percent of the logit spread.  Plain weights never clamp at
    the default scale."""
    cfg = _mid768()
    fr = make_frames(3, 3, cfg.image_size, 19)
    ids = torch.tensor([[101, 5, 9, 7], [101, 77, 3, 2], [101, 500, 41, 8]])
    # plain weights: nothing clamps at the default scale
    wp = quantize_weights_fp8(synthetic_weights(cfg, 0))
    mp = captioner_cls(cfg, wp, max_batch=3, max_text_len=8, weight_dtype="fp8_e4m3", compute="fp8_ffn")
    mp.forward_image_enc(fr)
    assert mp.fp8_saturations() == 0
    op = GitOracle(cfg, wp, emulate_bf16=True, emulate_fp8_act="ffn")
    with torch.no_grad():
        op.image_kv(op.forward_image_enc(fr)[1])
    assert op.f8_sat == 0
    # stress weights at the default scale: clamps, counted, the count equals the oracle's
    ws = quantize_weights_fp8(stress_weights(cfg, 0))
    m8 = captioner_cls(cfg, ws, max_batch=3, max_text_len=8, weight_dtype="fp8_e4m3", compute="fp8_ffn")
    assert m8.fp8_saturations() == 0
    _, v8 = m8.forward_image_enc(fr)
    n_dev = m8.fp8_saturations(reset=False)
    o8 = GitOracle(cfg, ws, emulate_bf16=True, emulate_fp8_act="ffn")
    with torch.no_grad():
        ov8, om8 = o8.forward_image_enc(fr)
        ikv8 = o8.image_kv(om8)
    n_orc = o8.f8_sat
    print(f"fp8_ffn on stress weights, scale 1/16: device clamped {n_dev} codes, the emulating oracle {n_orc}")
    assert n_dev > 0 and n_orc > 0
    assert abs(n_dev - n_orc) <= max(8, 0.02 * n_orc), (n_dev, n_orc)          # values within rounding noise of +-28 may fall either way
    assert m8.fp8_saturations() == n_dev and m8.fp8_saturations() == 0         # read + reset, then clean
    ob = GitOracle(cfg, ws, emulate_bf16=True)
    with torch.no_grad():
        ovb, omb = ob.forward_image_enc(fr)
        olb = ob.decoder_text(ob.image_kv(omb), ids)
        of = GitOracle(cfg, ws)
        olf = of.decoder_text(of.image_kv(of.forward_image_enc(fr)[1]), ids)
    spread = max(1.0, float(olb.std()) / 4.0)
    d_bf = float((olb - olf).abs().max())                   # what bf16 compute costs on these weights
    # sweep the scale: at every scale the mode either reports saturations or holds its accuracy -- never neither
    table, calibrated = [], None
    for e in range(-4, 1):
        scale = 2.0 ** e
        m8.set_fp8_scale(scale)
        _, v = m8.forward_image_enc(fr)
        n = m8.fp8_saturations()
        l = m8.forward_decoder(ids, v).cpu()
        oc = GitOracle(cfg, ws, emulate_bf16=True, emulate_fp8_act="ffn", fp8_scale=scale)
        with torch.no_grad():
            olc = oc.decoder_text(oc.image_kv(oc.forward_image_enc(fr)[1]), ids)
        row = dict(scale=scale, dev_sat=n, orc_sat=oc.f8_sat, d_own=float((l - olc).abs().max()), d_bf16=float((l - olb).abs().max()),
                   rms_own=_rms(l - olc), rms_bf16=_rms(l - olb), orc_d_bf16=float((olc - olb).abs().max()))
        table.append(row)
        print("fp8_ffn on stress weights:", {k: (round(x, 4) if isinstance(x, float) else x) for k, x in row.items()})
        assert abs(n - oc.f8_sat) <= max(8, 0.02 * oc.f8_sat), row
        if n == 0 and calibrated is None:
            calibrated = (scale, v, l, row)
    print(f"bf16 compute on the same weights: max |bf16-emulating - fp32 oracle| {d_bf:.3f}")
    assert table[0]["dev_sat"] > 0 and calibrated is not None
    scale, vc, lc, row = calibrated
    # What the sweep shows on these weights (GPUTEST log; docs/LAB_NOTEBOOK.md round 5): clamping is what wrecks the mode (rms
    # |dlogit| vs bf16 compute 0.69 at 1/16 with 264 clamped codes, 0.14 / 0.09 at 1/2 / 1 with none) and the counter is the
    # signal that tells the two apart.  Without clamping the mode still costs more here than on benign weights (max 0.5 - 1.1
    # against 0.17): e4m3 rounds a value of 60 in steps of 4 where bf16 rounds it in steps of 0.25, whatever the scale -- outlier
    # channels want bf16 compute, and the device and its oracle, rounding the same values at the same points, part by as much as
    # the mode itself costs when a value sits on a code boundary.
    for r in table:
        assert r["rms_own"] < 1.25 * r["rms_bf16"] + 0.01, r      # the device tracks its own oracle at least as closely as the mode costs
    assert table[0]["rms_bf16"] > 3.0 * row["rms_bf16"], (table[0], row)      # saturation is the failure; it is the counted one
    assert row["rms_bf16"] < 0.05 * 4.0 * spread, row               # no clamping: within 5 % of the logit spread in rms
    m8.set_fp8_scale(scale)
    # a handle created with the calibrated scale gives the same bits as one switched at run time
    m9 = captioner_cls(cfg, ws, max_batch=3, max_text_len=8, weight_dtype="fp8_e4m3", compute="fp8_ffn", fp8_scale=scale)
    _, v9 = m9.forward_image_enc(fr)
    assert torch.equal(v9, vc) and torch.equal(m9.forward_decoder(ids, v9).cpu(), lc)
    # argument checks: a power of two, fp8 compute only
    from gitcap._lib import GitcapError
    with pytest.raises(GitcapError, match="power of two"):
        m8.set_fp8_scale(0.3)
    with pytest.raises(ValueError, match="fp8_ffn"):
        captioner_cls(cfg, ws, max_batch=1, max_text_len=8, weight_dtype="fp8_e4m3", fp8_scale=0.25)
    mb = captioner_cls(cfg, ws, max_batch=3, max_text_len=8, weight_dtype="fp8_e4m3")
    mb.forward_image_enc(fr)
    assert mb.fp8_saturations() == 0                                           # bf16 compute: nothing is ever encoded


def test_config4_shape_on_stress_weights(captioner_cls):
    """BASELINE configs[4]'s shape (GIT-large, 10-frame clip, e4m3-valued weights, beam 4, 15 steps) on the stress family:
    teacher-forced logits against the bf16-emulating oracle (image half through image_kv), e4m3 storage bitwise equal to bf16
    storage of the same values, the device-resident search (synchronous and pipelined) bitwise equal to the host operator over
    the same kernels, and a clip searched alone equal to the same clip inside a batch."""
    from gitcap.config import git_large
    cfg = git_large(num_frames=10)
    wq = quantize_weights_fp8(stress_weights(cfg, 0))
    fr = make_frames(2, 10, cfg.image_size, 77)
    m = captioner_cls(cfg, wq, max_batch=2, max_frames=10, max_text_len=16, max_beams=4, weight_dtype="fp8_e4m3")
    emul, fp32 = GitOracle(cfg, wq, emulate_bf16=True), GitOracle(cfg, wq)
    ids = torch.tensor([[101, 2023, 2003, 1037, 3899, 2006]])
    _, vis = m.forward_image_enc(fr[:1])
    assert float(vis.abs().max()) > 10.0
    lg = m.forward_decoder(ids, vis).cpu()
    with torch.no_grad():
        l_e = emul.decoder_text(emul.image_kv(emul.forward_image_enc(fr[:1])[1]), ids)
        l_f = fp32.decoder_text(fp32.image_kv(fp32.forward_image_enc(fr[:1])[1]), ids)
    _check_logits("GIT-large stress", lg, l_e, l_f)
    mb = captioner_cls(cfg, wq, max_batch=2, max_frames=10, max_text_len=16, max_beams=4, weight_dtype="bf16")
    _, visb = mb.forward_image_enc(fr[:1])
    assert torch.equal(visb, vis) and torch.equal(mb.forward_decoder(ids, visb).cpu(), lg)
    dev = m.infer(fr, beam_size=4, max_steps=15, on_device=True)
    host = m.infer(fr, beam_size=4, max_steps=15, on_device=False)
    assert torch.equal(dev["predictions"], host["predictions"])
    assert torch.allclose(dev["logprobs"].cpu(), host["logprobs"].cpu(), atol=1e-5)
    futs = [m.infer_async(fr if i != 1 else fr[1:], beam_size=4, max_steps=15) for i in range(3)]      # CPU frames in -> CPU ids out
    res = [f.result() for f in futs]
    want = dev["predictions"].cpu()
    assert res[0]["predictions"].device.type == "cpu"
    assert torch.equal(res[0]["predictions"], want) and torch.equal(res[2]["predictions"], want)
    assert torch.equal(res[1]["predictions"][0], want[1])                              # a clip alone == inside the batch
    assert torch.equal(mb.infer(fr, beam_size=4, max_steps=15)["predictions"].cpu(), want)
