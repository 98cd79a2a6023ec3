"""Where does the stress family's end-to-end distance come from?  

tests/test_stress_gpu.py compares logits end to end and has to scale its tolerance with the conditioning of the input at hand.
Here every stage is checked ON ITS OWN: the bf16-emulating oracle is fed the DEVICE's input of that stage (ViT residual stream per
block: gitcap_dbg_enc_tap; decoder hidden states per layer: gitcap_hidden_states_*), so one stage's two computations are compared on
identical inputs and conditioning cannot amplify anything.  The tolerances are FIXED numbers, in bf16 ulps of the row maximum
(tests/stress_layers.py) -- nothing here is derived from a distance between oracles.  Reference stages: src/models/model.py:378
(image encoder), :412-418 (decoder over [image ; text]).

Also: the kernel-level tests of tests/test_kernels_gpu.py repeated on stress-SHAPED operands (outlier columns x 20-60, peaked
attention rows), and the fp8_ffn mode held to its own oracle at the single-layer level."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_layers import device_stages, format_table, stage_table, ulps_of_rowmax      # noqa: E402

from gitcap.config import GitCapConfig, git_base, git_large                               # noqa: E402
from gitcap.weights import quantize_weights_fp8, stress_weights, synthetic_weights        # noqa: E402
from oracle.git_oracle import GitOracle, make_frames                                      # noqa: E402

pytestmark = pytest.mark.gpu

# Fixed tolerances, in bf16 ulps of the row maximum, for ONE stage on shared inputs.  A stage's output differs between the device
# and the emulating oracle only where a GEMM operand sits within fp32 summation noise of a bf16 rounding boundary and falls the other
# way (one operand ulp, times a weight), and by the fp32 summation order itself.  Measured (profiles/r06_stress_divergence.md; plain
# and stress weights, GIT-base and the configs[4] shape): single-GEMM stages 0.0001; encoder blocks max 0.14 - 0.58, rms 0.008 - 0.048;
# decoder layers max 0.30 - 0.34 (plain) and 0.63 - 2.13 (stress: inside a layer a flipped q / k operand still moves a
# winner-take-all softmax row), rms 0.017 - 0.050 -- while END TO END the same runs reach 2.3 - 33 ulp.  The numbers below are those
# measurements with about 2 x headroom; none is computed from the data under test.
TOL_GEMM_STAGE_ULP = 0.01                   # patch embed + ln_pre, projection, text embedding, vocabulary head: one GEMM / gather + LayerNorm
TOL_ENC_MAX_ULP, TOL_DEC_MAX_ULP = 1.25, 4.0
TOL_RMS_ULP = 0.1


@pytest.fixture(scope="module")
def captioner_cls():
    from gitcap.model import GitCaptioner
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return GitCaptioner


def _check(title, rows):
    print(format_table(title, rows))
    bad = []
    for r in rows:
        mx, rms = r["shared"][0], r["shared"][1]
        tol = TOL_ENC_MAX_ULP if r["stage"].startswith("enc block") else TOL_DEC_MAX_ULP if r["stage"].startswith("dec layer") else TOL_GEMM_STAGE_ULP
        if not (mx <= tol and rms <= min(tol, TOL_RMS_ULP)):
            bad.append((r["stage"], round(mx, 4), round(rms, 4), tol))
    assert not bad, f"{title}: stages beyond the fixed single-stage tolerance (stage, max ulp, rms ulp, allowed max): {bad}"


@pytest.mark.parametrize("family", ["plain", "stress"])
def test_git_base_every_stage_on_device_inputs(captioner_cls, family):
    """GIT-base, 2 clips x 2 frames, T = 20 (the shape of tests/golden/hf_base_F2_stress.npz): 12 encoder blocks, projection,
    text embedding, 6 decoder layers over [image ; text], vocabulary head -- each against the emulating oracle on the device's
    own input of that stage."""
    cfg = git_base(2)
    w = (stress_weights if family == "stress" else synthetic_weights)(cfg, 0)
    fr = make_frames(2, 2, cfg.image_size, 1234)
    g = torch.Generator().manual_seed(20)
    ids = torch.randint(1000, cfg.vocab_size, (2, 20), generator=g)
    ids[:, 0] = cfg.cls_token_id
    m = captioner_cls(cfg, w, max_batch=2, max_frames=2, max_text_len=24)
    dev = device_stages(m, fr, ids)
    assert torch.isfinite(dev["enc"]).all() and torch.isfinite(dev["hidden"]).all()
    if family == "stress":
        assert float(dev["enc"].abs().max()) > 30.0 and float(dev["hidden"].abs().max()) > 15.0      # the outliers reach the device
    rows = stage_table(cfg, dev, GitOracle(cfg, w, emulate_bf16=True), fr, ids)
    _check(f"GIT-base, {family} weights, 2 clips x 2 frames, T = 20", rows)


def test_config4_shape_every_stage_on_device_inputs(captioner_cls):
    """BASELINE configs[4]'s shape on the stress family: GIT-large (24 blocks of width 1024, 257 tokens per frame), one 10-frame clip,
    e4m3-valued weights, T = 6."""
    cfg = git_large(num_frames=10)
    wq = quantize_weights_fp8(stress_weights(cfg, 0))
    fr = make_frames(1, 10, cfg.image_size, 77)
    ids = torch.tensor([[101, 2023, 2003, 1037, 3899, 2006]])
    m = captioner_cls(cfg, wq, max_batch=1, max_frames=10, max_text_len=16, weight_dtype="fp8_e4m3")
    dev = device_stages(m, fr, ids)
    rows = stage_table(cfg, dev, GitOracle(cfg, wq, emulate_bf16=True), fr, ids)
    _check("GIT-large (configs[4] shape), stress weights, 1 clip x 10 frames, T = 6", rows)


def _mid768(dec_layers=2):
    return GitCapConfig(image_size=64, patch_size=16, enc_width=768, enc_layers=2, enc_heads=12, enc_ffn=3072, dec_width=768,
                        dec_layers=dec_layers, dec_heads=12, dec_ffn=3072, vocab_size=997, max_text_pos=64, num_frames=3)


@pytest.mark.parametrize("scale", [1.0 / 16.0, 0.5])
def test_fp8_ffn_tracks_its_own_oracle_per_layer(captioner_cls, scale):
    """compute="fp8_ffn" at the single-layer level: FC1 -> GELU -> FC2 on e4m3 operands inside one block, device vs
    GitOracle(emulate_fp8_act="ffn") on the device's own block input.  The mode's own oracle must pin something: the device is
    at most HALF as far (rms) from it as from the bf16-compute oracle on the same input (= what the mode costs in that block) --
    the assertion tests/test_stress_gpu.py had to give up end to end.  At the default scale the stress weights clamp (counted);
    the oracle clamps the same codes, so the rule holds there too."""
    cfg = _mid768()
    ws = quantize_weights_fp8(stress_weights(cfg, 0))
    fr = make_frames(3, 3, cfg.image_size, 19)
    ids = torch.tensor([[101, 5, 9, 7], [101, 77, 3, 2], [101, 500, 41, 8]])
    m = captioner_cls(cfg, ws, max_batch=3, max_text_len=8, weight_dtype="fp8_e4m3", compute="fp8_ffn", fp8_scale=scale)
    dev = device_stages(m, fr, ids)
    own = GitOracle(cfg, ws, emulate_bf16=True, emulate_fp8_act="ffn", fp8_scale=scale)
    bf = GitOracle(cfg, ws, emulate_bf16=True)
    S_img = 3 * cfg.tokens_per_frame
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    rows = []
    with torch.no_grad():
        enc, hid = dev["enc"], dev["hidden"]
        for i in range(cfg.enc_layers):
            x_in = enc[i]
            if i + 1 < cfg.enc_layers:
                got, a, b = enc[i + 1], own.enc_block(i, x_in), bf.enc_block(i, x_in)
            else:
                got, a, b = dev["visual"], own.enc_post(own.enc_block(i, x_in), 3, 3), bf.enc_post(bf.enc_block(i, x_in), 3, 3)
            rows.append((f"enc block {i}", rms(got - a), rms(got - b), ulps_of_rowmax(got, a)[0]))
        for l in range(cfg.dec_layers):
            x_in, got = hid[:, l, :S_img], hid[:, l + 1, :S_img]
            rows.append((f"dec layer {l} (image rows)", rms(got - own.dec_layer_img(l, x_in)), rms(got - bf.dec_layer_img(l, x_in)),
                         ulps_of_rowmax(got, own.dec_layer_img(l, x_in))[0]))
    for r in rows:
        print(f"fp8_ffn scale {scale}: {r[0]}: rms device - own oracle {r[1]:.5f}, device - bf16-compute oracle {r[2]:.5f}, "
              f"max vs own {r[3]:.2f} ulp of row max")
    for name, r_own, r_mode, _ in rows:
        assert r_mode > 0 and r_own < 0.5 * r_mode, (name, r_own, r_mode)


# ------------------------------------------------------------------------------------------------ kernels, stress-shaped operands
def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K,post", [(1024, 768, 768, 0), (1024, 768, 3072, 0), (1024, 768, 3072, 1), (512, 1024, 1024, 1)])
def test_gemm_ln_outlier_operands(M, N, K, post):
    """gemm_ln (fused and unfused) on stress-shaped operands: four outlier columns x 20-60 in A (the FC2 operand behind saturating
    GELU units) and in the residual stream (values of +-60), outlier gammas x 20 -- against the fp64 torch reference.
    x (fp32): <= 0.02 ulp of the row maximum (fp32 accumulation of K products); LayerNorm output in bf16: <= 0.56 ulp (its own
    rounding is 0.5)."""
    from gitcap import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K + post)
    A = torch.randn(M, K, device="cuda", generator=g)
    A[:, torch.randperm(K, device="cuda", generator=g)[:4]] *= torch.tensor([20.0, 30.0, 45.0, 60.0], device="cuda")
    A = A.bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g)
    oc = torch.randperm(N, device="cuda", generator=g)[:4]
    resid[:, oc] *= 40.0
    gamma, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    gamma[oc] *= 20.0
    x = (A.double() @ W.double().t() + bias.double() + resid.double())
    ln = torch.nn.functional.layer_norm(x, (N,), gamma.double(), beta.double(), 1e-5)
    outs = []
    for fused, tile in ((1, 256), (0, 256), (0, 128)):
        of = torch.full((M, N), float("nan"), device="cuda")
        ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        assert lib.gitcap_dbg_gemm_ln(_p(A), _p(W), _p(bias), _p(resid), _p(gamma), _p(beta), ctypes.c_float(1e-5), _p(of), _p(ob),
                                      M, N, K, post, fused, tile, _stream()) == 0
        torch.cuda.synchronize()
        outs.append((of, ob))
    for of, ob in outs[1:]:
        assert torch.equal(outs[0][0], of) and torch.equal(outs[0][1], ob)
    of, ob = outs[0]
    e_f = ulps_of_rowmax(of.cpu(), (ln if post else x).cpu())
    e_b = ulps_of_rowmax(ob.float().cpu(), ln.cpu())
    print(f"gemm_ln outliers M={M} N={N} K={K} post={post}: fp32 out {e_f[0]:.4f} ulp (row max {e_f[2]:.0f}), bf16 LN out {e_b[0]:.3f} ulp (row max {e_b[2]:.0f})")
    assert e_f[2] > 50.0
    assert e_f[0] < 0.02 and e_b[0] < 0.56


@pytest.mark.parametrize("G,S,H", [(2, 394, 12), (1, 1182, 12), (2, 514, 16)])
def test_attn_full_peaked_scores(G, S, H):
    """attn_full on stress-shaped q|k|v: head 0's q and k x 3 (scores x 9: rows dominated by one or two keys, running maxima that jump
    by tens of units between key tiles -> the online-softmax rescale), one key that dominates EVERY row of head 1, and v with an
    outlier column (x 40, inside the peaked head).  Against the fp64 reference of the same bf16 operands (P enters P.V as bf16).
    The bound is the rounding the arithmetic is DEFINED to have, element by element: both sides round every probability to bf16 once
    (the device relative to the running maximum of the key tile, the reference relative to the row maximum: two roundings of at most
    2^-9 each, i.e. 2^-8 of the TERM p_j |v_jd|) and the device rounds its output to bf16 (2^-9 of the value).  Winner-take-all rows
    over outlier values of both signs cancel, so an error bound in units of the OUTPUT would be unbounded (measured: up to 3.9 ulp of
    the row maximum); in units of the terms it is attained: measured 0.92 - 1.012 of the first-order bound over the three shapes (a
    maximum over ~10^6 elements whose one or two dominant terms round in opposite directions on the two sides).  Allowed: 1.1 x, the
    10 % for what the first-order bound leaves out (the output rounding acts on the device's own value, exp2 / reciprocal are 1-ulp
    approximations, fp32 accumulation)."""
    from gitcap import _lib
    lib = _lib.load()
    W = H * 64
    g = torch.Generator(device="cuda").manual_seed(S + H)
    qkv = torch.randn(G * S, 3 * W, device="cuda", generator=g) * 1.5
    qkv[:, 0:64] *= 3.0
    qkv[:, W:W + 64] *= 3.0
    qkv[:, 64:128] += 1.0                                                      # every query of head 1 leans the same way ...
    qkv[S // 2::S, W + 64:W + 128] = 4.0                                       # ... towards ONE key per group (score ~ 32 above the rest)
    qkv[:, 2 * W + 7] *= 40.0                                                  # outlier value column
    qkv = qkv.bfloat16()
    ctx = torch.zeros(G * S, W, device="cuda", dtype=torch.bfloat16)
    assert lib.gitcap_dbg_attn_full(_p(qkv), _p(ctx), G, S, H, _stream()) == 0
    q, k, v = (t.double().view(G, S, H, 64).transpose(1, 2) for t in qkv.split(W, dim=1))
    s = q @ k.transpose(-1, -2) * 0.125
    p = torch.exp(s - s.max(-1, keepdim=True).values)
    l = p.sum(-1, keepdim=True)
    ref = (p.bfloat16().double() @ v) / l
    terms = (p @ v.abs()) / l                                                  # sum_j p_j |v_jd| / l: the magnitude the roundings act on
    bound = 2.0 ** -8 * terms + 2.0 ** -9 * ref.abs() + 1e-6
    got = ctx.double().view(G, S, H, 64).transpose(1, 2)
    ratio = ((got - ref).abs() / bound)
    e = ulps_of_rowmax(got.transpose(1, 2).reshape(G * S, H, 64).cpu(), ref.transpose(1, 2).reshape(G * S, H, 64).cpu())
    print(f"attn_full peaked G={G} S={S} H={H}: max error / rounding bound {float(ratio.max()):.3f} (head 0: {float(ratio[:, 0].max()):.3f}, "
          f"head 1: {float(ratio[:, 1].max()):.3f}); in ulps of the (row, head) maximum {e[0]:.2f} (rms {e[1]:.3f}), largest |ctx| {e[2]:.0f}")
    assert float(s[:, 0].amax(-1).mean()) > 20.0            # the scores really are peaked
    assert float(ratio.max()) < 1.1
    assert e[1] < 0.3


def test_text_rows_one_layer_peaked_head(captioner_cls):
    """txt_block / ffn_txt / ln_reduce through gitcap_text_forward on a ONE-layer decoder whose head 0 is peaked harder than the
    stress default (q, k x 4: scores x 16 -- winner-take-all rows): the text rows of that layer against the emulating oracle on the
    device's own layer input, at T = 1 ... 20 and with 1, 2 and 5 rows (row-prologue and row-kernel forms)."""
    cfg = _mid768(dec_layers=1)
    w = stress_weights(cfg, 0, qk_gain=4.0)
    emul = GitOracle(cfg, w, emulate_bf16=True)
    S_img = 3 * cfg.tokens_per_frame
    g = torch.Generator().manual_seed(5)
    for rows_n in (1, 2, 5):
        m = captioner_cls(cfg, w, max_batch=rows_n, max_text_len=24)
        fr = make_frames(rows_n, 3, cfg.image_size, 50 + rows_n)
        ids = torch.randint(5, cfg.vocab_size, (rows_n, 20), generator=g)
        ids[:, 0] = cfg.cls_token_id
        dev = device_stages(m, fr, ids)
        with torch.no_grad():
            ref = emul.dec_layer_full(0, dev["hidden"][:, 0], S_img)
        e_txt = ulps_of_rowmax(dev["hidden"][:, 1, S_img:], ref[:, S_img:])
        e_img = ulps_of_rowmax(dev["hidden"][:, 1, :S_img], ref[:, :S_img])
        print(f"one peaked layer, {rows_n} rows: text rows {e_txt[0]:.3f} ulp (rms {e_txt[1]:.4f}), image rows {e_img[0]:.3f} ulp (rms {e_img[1]:.4f})")
        assert e_txt[0] <= TOL_DEC_MAX_ULP and e_txt[1] <= TOL_RMS_ULP
        assert e_img[0] <= TOL_DEC_MAX_ULP and e_img[1] <= TOL_RMS_ULP
