"""Per-kernel numerics on the MI355X: each hand-written kernel, run alone through the C-ABI test
hooks (gitcap_dbg_*), against a plain PyTorch fp32 reference of the same op on the SAME bf16
inputs.  This is where the 1e-3 tolerance lives: with identical operands only the summation order
differs."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from gitcap import _lib
    assert torch.cuda.is_available()
    return _lib.load()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("tile", [64, 128, 256])
@pytest.mark.parametrize("M,N,K,epi", [(512, 768, 768, 0), (256, 256, 64, 4), (768, 2304, 768, 0),
                                       (512, 3072, 768, 1), (512, 3072, 768, 2), (512, 768, 3072, 3),
                                       (256, 1536, 768, 0), (1024, 1024, 1024, 3)])
def test_gemm_vs_fp32_reference(lib, tile, M, N, K, epi):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    # asymmetric operands + a bias that depends on n catch transposed / shifted epilogues
    bias = torch.linspace(-1, 1, N, device="cuda")
    resid = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    rc = lib.gitcap_dbg_gemm(_p(A), _p(W), _p(bias), _p(resid), _p(out), M, N, K, epi, tile, _stream())
    assert rc == 0
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif epi == 2:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 3:
        ref = ref + resid
    if out.dtype == torch.bfloat16:      # output rounding: half a bf16 ulp (2^-9 relative) on top of 1e-3
        assert torch.allclose(out.float(), ref, rtol=2 ** -8, atol=2e-3)
    else:
        assert torch.allclose(out, ref, rtol=1e-3, atol=1e-3)


def test_gemm_identity_weight_asymmetric_input(lib):
    """A = anything, W = I: the output must be A itself, exactly (catches row/col swaps)."""
    M = N = K = 256
    A = torch.arange(M * K, device="cuda", dtype=torch.float32).reshape(M, K).remainder(251).bfloat16()
    W = torch.eye(N, K, device="cuda").bfloat16()
    for tile in (64, 128, 256):
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        assert lib.gitcap_dbg_gemm(_p(A), _p(W), None, None, _p(out), M, N, K, 4, tile, _stream()) == 0
        assert torch.equal(out, A.float())


def test_gemm_tile_variants_are_bitwise_equal(lib):
    """Every tile kernel accumulates each output over ascending k with the same MFMA, so the 128x128 and 64x64
    kernels must reproduce the 256x256 product kernel bit for bit (this is what makes results independent of which
    kernel a shape is routed to)."""
    M, N, K = 1024, 768, 768
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g)
    for epi, dt in ((0, torch.bfloat16), (1, torch.bfloat16), (3, torch.float32)):
        outs = []
        for tile in (256, 128, 64):
            out = torch.empty(M, N, device="cuda", dtype=dt)
            assert lib.gitcap_dbg_gemm(_p(A), _p(W), _p(bias), _p(resid), _p(out), M, N, K, epi, tile, _stream()) == 0
            outs.append(out)
        for o in outs[1:]:
            assert torch.equal(outs[0], o)


@pytest.mark.parametrize("M,N,K,post,with_resid", [(1024, 768, 768, 0, True), (1024, 768, 768, 1, True), (768, 768, 3072, 1, False),
                                                   (512, 1024, 256, 0, True), (19200, 768, 768, 0, True), (19200, 768, 3072, 1, True),
                                                   (75776, 768, 768, 0, True), (2304, 1024, 1024, 1, True)])
def test_gemm_layernorm_epilogue_equals_gemm_then_layernorm(lib, M, N, K, post, with_resid):
    """The residual GEMM that normalises its own rows (tiles of a 256-row block exchange segment statistics) must give the
    bits of the GEMM (either tile kernel) followed by the row kernel -- which of them runs depends on the batch size --
    and both must be LayerNorm(A W^T + bias + resid) to fp32 accuracy.  19200 rows = the bench shape (225 tiles, one round);
    75776 rows = 888 tiles, several rounds on 256 CUs: the tiles of a row block must stay consecutive workgroups of one XCD
    (a tile that waits for a sibling which is queued behind it would spin until the trap); 2304 x 1024 = 9 row blocks of 4
    tiles, not a multiple of the 8 XCDs (padded grid, surplus workgroups leave at once)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g) * 2 + 0.5 if with_resid else None
    gamma, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    outs = []
    for fused, tile in ((1, 256), (0, 256), (0, 128), (0, 64)):
        of = torch.full((M, N), float("nan"), device="cuda")
        ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(2):          # twice: the exchange barrier must be reusable
            assert lib.gitcap_dbg_gemm_ln(_p(A), _p(W), _p(bias), _p(resid), _p(gamma), _p(beta), ctypes.c_float(1e-5), _p(of), _p(ob),
                                          M, N, K, post, fused, tile, _stream()) == 0
        torch.cuda.synchronize()
        outs.append((of, ob))
    for of, ob in outs[1:]:
        assert torch.equal(outs[0][0], of) and torch.equal(outs[0][1], ob)
    x = A.float() @ W.float().t() + bias + (resid if with_resid else 0)
    ln = torch.nn.functional.layer_norm(x, (N,), gamma, beta, 1e-5)
    of, ob = outs[0]
    assert torch.allclose(of, ln if post else x, rtol=1e-4, atol=2e-4 * K ** 0.5)
    assert (ob.float() - ln).abs().max().item() < 0.05


@pytest.mark.parametrize("G,S,H", [(3, 197, 12), (2, 1182, 12), (4, 17, 2), (1, 64, 1), (2, 65, 3), (1, 257, 16)])
def test_attn_full_vs_reference(lib, G, S, H):
    W = H * 64
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = (torch.randn(G * S, 3 * W, device="cuda", generator=g) * 1.5).bfloat16()
    ctx = torch.zeros(G * S, W, device="cuda", dtype=torch.bfloat16)
    assert lib.gitcap_dbg_attn_full(_p(qkv), _p(ctx), G, S, H, _stream()) == 0
    q, k, v = (t.float().view(G, S, H, 64).transpose(1, 2) for t in qkv.split(W, dim=1))
    s = q @ k.transpose(-1, -2) * 0.125
    p = torch.exp(s - s.max(-1, keepdim=True).values)
    ref = (p.bfloat16().float() @ v) / p.sum(-1, keepdim=True)          # P enters P.V as bf16
    ref = ref.transpose(1, 2).reshape(G * S, W)
    err = (ctx.float() - ref).abs().max().item()
    assert err < 2e-2, err           # |ctx| ~ 1; bf16 output ulp 4e-3..8e-3 + P rounded at a different scale


def test_attn_full_forced_rescale(lib):
    """Online-softmax rescale branch: one late key dominates every row (max jumps in the last tile)."""
    G, S, H = 1, 200, 1
    qkv = torch.zeros(S, 192, device="cuda")
    qkv[:, 0:64] = 0.5
    qkv[:, 64:128] = torch.randn(S, 64, device="cuda") * 0.1
    qkv[S - 3, 64:128] = 4.0                                              # spike in the last 64-key tile
    qkv[:, 128:192] = torch.arange(S, device="cuda", dtype=torch.float32)[:, None] / S
    qkv = qkv.bfloat16()
    ctx = torch.zeros(S, 64, device="cuda", dtype=torch.bfloat16)
    assert lib.gitcap_dbg_attn_full(_p(qkv), _p(ctx), G, S, H, _stream()) == 0
    q, k, v = (t.float() for t in qkv.split(64, dim=1))
    ref = torch.softmax(q @ k.t() * 0.125, -1) @ v
    assert (ctx.float() - ref).abs().max().item() < 1e-2


@pytest.mark.parametrize("rows,D", [(1000, 768), (77, 1024), (33, 128)])
def test_layernorm_vs_reference(lib, rows, D):
    x = torch.randn(rows, D, device="cuda") * 3 + 1
    gamma, beta = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    of = torch.empty_like(x)
    ob = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    assert lib.gitcap_dbg_layernorm(_p(x), _p(gamma), _p(beta), ctypes.c_float(1e-5), rows, D, _p(of), _p(ob), _stream()) == 0
    ref = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    assert torch.allclose(of, ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(ob, of.bfloat16())


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 1024, 1024), (768, 4096, 1024), (512, 1024, 4096), (256, 768, 3072)])
def test_gemm_fp8_vs_fp32_reference(lib, M, N, K):
    """gemm_f8.hip (v_mfma_f32_16x16x128_f8f6f4): e4m3 codes in, exact products, fp32 accumulation -- against the fp32
    matmul of the dequantised operands; then the e4m3-output GELU epilogues against the same reference rounded to e4m3."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a_f = (torch.randn(M, K, device="cuda", generator=g) * 1.5).clamp(-27, 27)
    w_f = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    ascale = 1.0 / 16.0
    A8 = (a_f / ascale).to(torch.float8_e4m3fn)
    wscale = torch.exp2(torch.ceil(torch.log2(w_f.abs().amax(dim=1) / 448.0)))          # power of two per row
    W8 = (w_f / wscale[:, None]).to(torch.float8_e4m3fn)
    bias = torch.linspace(-1, 1, N, device="cuda")
    ref = (A8.float() * ascale) @ (W8.float() * wscale[:, None]).t() + bias
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    rc = lib.gitcap_dbg_gemm_f8(_p(A8), _p(W8), _p(wscale), ascale, _p(bias), _p(out), M, N, K, 4, 0.0, _stream())
    assert rc == 0
    assert torch.allclose(out, ref, rtol=1e-3, atol=1e-3), float((out - ref).abs().max())
    for epi, act in ((8, lambda x: x * torch.sigmoid(1.702 * x)), (9, torch.nn.functional.gelu)):
        o8 = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        assert lib.gitcap_dbg_gemm_f8(_p(A8), _p(W8), _p(wscale), ascale, _p(bias), _p(o8), M, N, K, epi, 16.0, _stream()) == 0
        got = o8.view(torch.float8_e4m3fn).float() / 16.0
        want = (act(ref) * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 16.0
        # a value on an e4m3 rounding boundary may fall either way (the GELU differs in the last bits): one e4m3 step
        step = torch.maximum(want.abs() * 2 ** -3, torch.full_like(want, 2 ** -9 / 16.0 * 8))
        assert bool(((got - want).abs() <= step + 1e-6).all()), float(((got - want).abs() / step).max())
        assert float((got != want).float().mean()) < 0.02
