"""Student caption decoder (SURVEY.md par. 8 row f.2).

CPU part: pins oracle/student_oracle.py to the fixtures oracle/gen_golden_student.py generated from
``torch.nn.TransformerDecoder`` (the module the reference instantiates, model.py:82-85) called with the
reference's mask helpers.  GPU part: the HIP path through the C ABI against that oracle."""
import os

import numpy as np
import pytest
import torch

from gitcap.student_config import (student_base, student_shapes, student_stress_weights, student_synthetic_weights,
                                   student_tiny, positional_table)
from oracle.student_oracle import StudentOracle, make_memory


def test_tiny_oracle_matches_torch_decoder(golden_dir):
    cfg = student_tiny()
    g = np.load(os.path.join(golden_dir, "student_tiny.npz"))
    orc = StudentOracle(cfg, student_synthetic_weights(cfg, 0))
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    logits = orc.forward_decoder(torch.from_numpy(g["y"]), mem)
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-5
    assert np.array_equal(orc.greedy_decode(mem, 12, stop="never").numpy(), g["greedy_ids"])


def test_oracle_beam_search_matches_loop_restatement(golden_dir):
    """model.py:189-318: the oracle's vectorised beam search vs the loop-for-loop restatement over the
    torch modules (oracle/gen_golden_student.py)."""
    cfg = student_tiny()
    g = np.load(os.path.join(golden_dir, "student_tiny.npz"))
    orc = StudentOracle(cfg, student_synthetic_weights(cfg, 0))
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    assert np.array_equal(orc.beam_search(mem, 9, 3).numpy(), g["beam_k3"]) and g["beam_k3"].shape == (3, 9)
    assert np.array_equal(orc.beam_search(mem, 6, 4).numpy(), g["beam_k4"])
    # beam 1 degenerates to greedy
    assert np.array_equal(orc.beam_search(mem, 9, 1).numpy(), orc.greedy_decode(mem, 8, stop="never").numpy())


def test_pad_tokens_are_masked_as_keys(golden_dir):
    """model.py:134 + masking.py:14: a generated PAD (id 0) is never attended to afterwards."""
    cfg = student_tiny()
    g = np.load(os.path.join(golden_dir, "student_tiny_pad.npz"))
    w = student_synthetic_weights(cfg, 0)
    w["linear.bias"] = w["linear.bias"].copy()
    w["linear.bias"][cfg.pad_token_id] = 50.0
    orc = StudentOracle(cfg, w)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))[:1]
    logits = orc.forward_decoder(torch.from_numpy(g["y"]), mem)
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-5
    assert np.array_equal(orc.greedy_decode(mem, 6, stop="never").numpy(), g["greedy_ids"])


def test_base_oracle_matches_torch_decoder(golden_dir):
    cfg = student_base()
    g = np.load(os.path.join(golden_dir, "student_base.npz"))
    orc = StudentOracle(cfg, student_synthetic_weights(cfg, 0))
    mem = make_memory(2, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    ids = torch.from_numpy(g["greedy_ids"])
    logits = orc.forward_decoder(ids[:, :-1], mem)                  # teacher-forced: a near-tie cannot cascade
    top = torch.gather(logits, 2, torch.from_numpy(g["top_ids"]))
    assert np.abs(top.numpy() - g["top_vals"]).max() < 1e-4
    assert np.abs(logits[:, :, :16].numpy() - g["first16"]).max() < 1e-4
    margin = g["top_vals"][..., 0] - g["top_vals"][..., 1]
    assert np.array_equal(logits.argmax(-1).numpy()[margin > 1e-3], ids[:, 1:].numpy()[margin > 1e-3])


@pytest.mark.parametrize("name", ["tiny", "base"])
def test_oracle_matches_torch_decoder_on_stress_weights(golden_dir, name):
    """Second weight family (student_stress_weights: outlier LayerNorm channels, big ReLU inputs, a peaked head) against the
    torch.nn.TransformerDecoder goldens of oracle/gen_golden_student.py: teacher-forced on the golden ids, top-8 logits per step."""
    cfg = student_tiny() if name == "tiny" else student_base()
    g = np.load(os.path.join(golden_dir, f"student_{name}_stress.npz"))
    orc = StudentOracle(cfg, student_stress_weights(cfg, 0))
    B = g["greedy_ids"].shape[0]
    mem = make_memory(B, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    ids = torch.from_numpy(g["greedy_ids"])
    logits = orc.forward_decoder(ids[:, :-1], mem)
    top = torch.gather(logits, 2, torch.from_numpy(g["top_ids"]))
    assert np.abs(top.numpy() - g["top_vals"]).max() < 3e-4
    assert np.abs(logits[:, :, :16].numpy() - g["first16"]).max() < 3e-4
    margin = g["top_vals"][..., 0] - g["top_vals"][..., 1]
    assert np.array_equal(logits.argmax(-1).numpy()[margin > 2e-3], ids[:, 1:].numpy()[margin > 2e-3])


def test_stop_rule_and_shapes():
    cfg = student_tiny()
    w = student_synthetic_weights(cfg, 0)
    assert set(w) == set(student_shapes(cfg)) and w["pos_enc.pe"].shape == (1, cfg.max_pos, cfg.d_model)
    assert np.array_equal(w["pos_enc.pe"], positional_table(cfg.d_model, cfg.max_pos))
    w["linear.bias"] = w["linear.bias"].copy()
    w["linear.bias"][cfg.sep_token_id] = 1e4
    orc = StudentOracle(cfg, w)
    mem = make_memory(2, cfg.mem_tokens, cfg.d_model, 3)
    ids = orc.greedy_decode(mem, 8)                                  # model.py:184: all rows SEP in the same step
    assert ids.shape == (2, 2) and bool((ids[:, 1] == cfg.sep_token_id).all())
    assert orc.greedy_decode(mem, 8, stop="never").shape == (2, 9)


def test_bf16_emulation_is_close():
    cfg = student_tiny()
    w = student_synthetic_weights(cfg, 0)
    mem = make_memory(2, cfg.mem_tokens, cfg.d_model, 5)
    y = torch.tensor([[1, 5, 9, 33], [1, 77, 0, 15]])
    a = StudentOracle(cfg, w).forward_decoder(y, mem)
    b = StudentOracle(cfg, w, emulate_bf16=True).forward_decoder(y, mem)
    err = (a - b).abs().max().item()
    assert 1e-4 < err < 0.08 * a.std().item(), err


# ---------------------------------------------------------------------------------------------------
# GPU: HIP path (gitcap/student.py -> C ABI -> csrc/student.hip) against the oracle
# ---------------------------------------------------------------------------------------------------
TOL_EMU = 0.03        # max |logit| error vs the bf16-emulating oracle (logit std ~1)
TOL_F32 = 0.10        # vs the fp32 oracle / the torch.nn goldens
NEAR_TIE = 0.15       # token parity is asserted where the oracle's top-1 margin exceeds this


def _student(cfg, w, **kw):
    from gitcap.student import StudentCaptioner
    return StudentCaptioner(cfg=cfg, weights=w, device="cuda:0", **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base"])
def test_gpu_forward_decoder_matches_oracle(name):
    cfg = student_tiny() if name == "tiny" else student_base()
    w = student_synthetic_weights(cfg, 0)
    m = _student(cfg, w, max_batch=4, max_text_len=12)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, 21)
    V = cfg.vocab_length
    g = torch.Generator().manual_seed(5)
    y = torch.randint(1, V, (3, 9), generator=g)
    y[:, 0] = cfg.cls_token_id
    y[1, 3] = 0; y[1, 5] = 0; y[2, 6:] = 0                      # PAD keys (masking.py:14)
    got = m.forward_decoder(y, mem).cpu()
    emu = StudentOracle(cfg, w, emulate_bf16=True).forward_decoder(y, mem)
    f32 = StudentOracle(cfg, w).forward_decoder(y, mem)
    assert torch.isfinite(got).all()
    assert (got - emu).abs().max().item() < TOL_EMU * max(1.0, emu.std().item())
    assert (got - f32).abs().max().item() < TOL_F32 * max(1.0, f32.std().item())


@pytest.mark.gpu
def test_gpu_tiny_matches_torch_goldens(golden_dir):
    cfg = student_tiny()
    g = np.load(os.path.join(golden_dir, "student_tiny.npz"))
    m = _student(cfg, student_synthetic_weights(cfg, 0), max_batch=4, max_text_len=12)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    got = m.forward_decoder(torch.from_numpy(g["y"]), mem).cpu().numpy()
    assert np.abs(got - g["logits"]).max() < TOL_F32 * max(1.0, g["logits"].std())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base"])
def test_gpu_greedy_kv_cache_and_token_parity(name, golden_dir):
    cfg = student_tiny() if name == "tiny" else student_base()
    w = student_synthetic_weights(cfg, 0)
    m = _student(cfg, w, max_batch=4, max_text_len=25)
    seed = 11 if name == "tiny" else 12
    B, L = (3, 12) if name == "tiny" else (2, 25)
    mem = make_memory(B, cfg.mem_tokens, cfg.d_model, seed)
    ids = m.greedy_decode(mem, max_len=L, stop="never")
    assert ids.shape == (B, L + 1) and bool((ids[:, 0] == cfg.cls_token_id).all()) and ids.device.type == "cpu"
    # exact KV cache: the cached token loop == one full teacher-forced pass over its own output
    full = m.forward_decoder(ids[:, :-1], mem).cpu()
    assert torch.equal(full.argmax(-1), ids[:, 1:])
    # token parity with the oracle wherever the oracle's decision is not a near-tie
    emu = StudentOracle(cfg, w, emulate_bf16=True).forward_decoder(ids[:, :-1], mem)
    top2 = emu.topk(2, dim=-1).values
    sure = (top2[..., 0] - top2[..., 1]) > NEAR_TIE
    assert sure.float().mean().item() > 0.3            # the comparison below is not vacuous
    assert torch.equal(emu.argmax(-1)[sure], ids[:, 1:][sure])
    assert (full - emu).abs().max().item() < TOL_EMU * max(1.0, emu.std().item())
    # and with the torch.nn.TransformerDecoder goldens (same seeds) on the confident steps
    g = np.load(os.path.join(golden_dir, f"student_{name}.npz"))
    gold = torch.from_numpy(g["greedy_ids"])
    same_prefix = (ids[:, :gold.shape[1]] == gold).long().cumprod(dim=1).sum(dim=1)
    if name == "base":
        margin = torch.from_numpy(g["top_vals"][..., 0] - g["top_vals"][..., 1])
        for b in range(B):
            k = int(same_prefix[b])                 # first divergence, if any, must sit on a near-tie of the golden run
            assert k == gold.shape[1] or margin[b, k - 1] < NEAR_TIE, (b, k, float(margin[b, k - 1]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base"])
def test_gpu_student_on_stress_weights(name, golden_dir):
    """The student decoder on the stress family: exact KV cache (cached loop == one teacher-forced pass, bitwise), logits against
    the bf16-emulating oracle (tolerance scaled by what the rounding points cost on this input, as in tests/test_stress_gpu.py)
    and the torch goldens on the confident steps; beam search device == host form."""
    cfg = student_tiny() if name == "tiny" else student_base()
    w = student_stress_weights(cfg, 0)
    g = np.load(os.path.join(golden_dir, f"student_{name}_stress.npz"))
    B, L = g["greedy_ids"].shape[0], g["greedy_ids"].shape[1] - 1
    m = _student(cfg, w, max_batch=3 * B, max_text_len=L)
    mem = make_memory(B, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    ids = m.greedy_decode(mem, max_len=L, stop="never")
    full = m.forward_decoder(ids[:, :-1], mem).cpu()
    assert torch.equal(full.argmax(-1), ids[:, 1:])
    emu = StudentOracle(cfg, w, emulate_bf16=True).forward_decoder(ids[:, :-1], mem)
    f32 = StudentOracle(cfg, w).forward_decoder(ids[:, :-1], mem)
    d = float((emu - f32).abs().max())
    tol = max(TOL_EMU * max(1.0, emu.std().item()), 1.5 * d)
    print(f"student {name} stress: |emul - fp32| {d:.3f}, device vs emul {float((full - emu).abs().max()):.3f} (tol {tol:.3f})")
    assert (full - emu).abs().max().item() < tol
    gold = torch.from_numpy(g["greedy_ids"])
    margin = torch.from_numpy(g["top_vals"][..., 0] - g["top_vals"][..., 1])
    same_prefix = (ids == gold).long().cumprod(dim=1).sum(dim=1)
    for b in range(B):
        k = int(same_prefix[b])
        assert k == gold.shape[1] or margin[b, k - 1] < max(NEAR_TIE, 2 * tol), (b, k, float(margin[b, k - 1]))
    assert torch.equal(m.beam_search(mem, max_len=min(L, 12), k=3), m.beam_search_host(mem, max_len=min(L, 12), k=3))


@pytest.mark.gpu
def test_gpu_generated_pad_is_masked(golden_dir):
    cfg = student_tiny()
    g = np.load(os.path.join(golden_dir, "student_tiny_pad.npz"))
    w = student_synthetic_weights(cfg, 0)
    w["linear.bias"] = w["linear.bias"].copy()
    w["linear.bias"][cfg.pad_token_id] = 50.0
    m = _student(cfg, w, max_batch=2, max_text_len=8)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))[:1]
    assert np.array_equal(m.greedy_decode(mem, max_len=6, stop="never").numpy(), g["greedy_ids"])
    got = m.forward_decoder(torch.from_numpy(g["y"]), mem).cpu().numpy()
    assert np.abs(got - g["logits"]).max() < TOL_F32 * max(1.0, g["logits"].std())


@pytest.mark.gpu
def test_gpu_stop_rule_pickle_and_errors():
    import pickle
    from gitcap._lib import GitcapError
    cfg = student_tiny()
    w = student_synthetic_weights(cfg, 0)
    w["linear.bias"] = w["linear.bias"].copy()
    w["linear.bias"][cfg.sep_token_id] = 1e4
    m = _student(cfg, w, max_batch=2, max_text_len=8)
    mem = make_memory(2, cfg.mem_tokens, cfg.d_model, 3)
    ids = m.greedy_decode(mem.cuda(), max_len=8)                    # model.py:184
    assert ids.shape == (2, 2) and ids.device.type == "cuda" and bool((ids[:, 1] == cfg.sep_token_id).all())
    assert m.greedy_decode(mem, max_len=8, stop="never").shape == (2, 9)
    m2 = pickle.loads(pickle.dumps(m))
    assert torch.equal(m2.greedy_decode(mem, max_len=8, stop="never"), m.greedy_decode(mem, max_len=8, stop="never"))
    assert set(m.state_dict()) == set(student_shapes(cfg))
    with pytest.raises(ValueError):
        m.greedy_decode(mem, max_len=9)                             # > max_text_len
    with pytest.raises(ValueError):
        m.greedy_decode(make_memory(3, cfg.mem_tokens, cfg.d_model, 3), max_len=4)   # > max_batch
    with pytest.raises(ValueError):
        m.forward_decoder(torch.ones(2, 3, dtype=torch.long), mem[:, :3])            # wrong memory shape
    with pytest.raises(GitcapError):
        m.greedy_decode(torch.zeros(2, 6, 3, 32, 32), max_len=4)   # frames without an image_encoder
    with pytest.raises(GitcapError):
        m.to("cpu")
    from gitcap.student import StudentCaptioner
    with pytest.raises(GitcapError, match="d_model"):
        StudentCaptioner(d_model=512, n_head=8, d_ffn=1024)             # no kernel instantiation: refused at create
    sd = {k: v for k, v in m.state_dict().items() if k != "pos_enc.pe"}
    sd["image_encoder.model.stem.weight"] = torch.zeros(1)          # foreign keys of a reference checkpoint are ignored
    m.load_state_dict(sd)                                           # pos_enc.pe rebuilt from the formula
    assert m.greedy_decode(mem, max_len=8, stop="never").shape == (2, 9)


@pytest.mark.gpu
def test_gpu_beam_search(golden_dir):
    """k beams as rows of the HIP forward_decoder; compared with the bf16-emulating oracle step by step
    (teacher-forced on the device's own beams, so a near-tie cannot cascade) and with the goldens."""
    cfg = student_tiny()
    w = student_synthetic_weights(cfg, 0)
    g = np.load(os.path.join(golden_dir, "student_tiny.npz"))
    m = _student(cfg, w, max_batch=12, max_text_len=12)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, int(g["mem_seed"]))
    got = m.beam_search(mem, max_len=9, k=3)
    assert got.shape == (3, 9) and bool((got[:, 0] == cfg.cls_token_id).all())
    assert torch.equal(m.beam_search(mem, max_len=9, k=1), m.greedy_decode(mem, max_len=8, stop="never"))
    emu = StudentOracle(cfg, w, emulate_bf16=True)
    want = emu.beam_search(mem, 9, 3)
    # sequence log-probability of both winners under the oracle: the device's choice must be as good as the
    # oracle's up to bf16 noise (identical sequences in the common case)
    def seq_logp(ids):
        lp = torch.log_softmax(emu.forward_decoder(ids[:, :-1], mem), dim=-1)
        return lp.gather(2, ids[:, 1:].unsqueeze(-1)).squeeze(-1).sum(dim=1)
    assert (seq_logp(got) - seq_logp(want)).abs().max().item() < 0.25
    assert (got == torch.from_numpy(g["beam_k3"])).float().mean().item() > 0.8
    with pytest.raises(ValueError):
        m.beam_search(make_memory(5, cfg.mem_tokens, cfg.d_model, 1), max_len=5, k=3)      # 15 rows > max_batch
    # the device-resident search (KV-cached, candidates ranked and beams reordered on the GPU) against the same search
    # driven from the host over full-prefix forward_decoder calls: the cache is exact (bitwise), so the same beams
    for kk, ml in ((3, 9), (4, 6), (2, 12), (1, 5)):
        assert torch.equal(m.beam_search(mem, max_len=ml, k=kk), m.beam_search_host(mem, max_len=ml, k=kk)), (kk, ml)
    # and at the reference's model size (config.py:78-83), where a greedy call in between must still see its own cache
    cfgb = student_base()
    mb = _student(cfgb, student_synthetic_weights(cfgb, 0), max_batch=8, max_text_len=25)
    memb = make_memory(2, cfgb.mem_tokens, cfgb.d_model, 5)
    gr = mb.greedy_decode(memb, max_len=10, stop="never")
    bs = mb.beam_search(memb, max_len=12, k=4)
    assert torch.equal(bs, mb.beam_search_host(memb, max_len=12, k=4))
    assert torch.equal(mb.greedy_decode(memb, max_len=10, stop="never"), gr)


@pytest.mark.gpu
def test_gpu_student_batch_invariance_and_determinism():
    """Every kernel on the path sums in an order that depends only on the row itself: a clip must decode to
    bit-identical logits and ids alone, inside a batch of 5, and on a repeated call (graph replay included)."""
    cfg = student_base()
    w = student_synthetic_weights(cfg, 0)
    m = _student(cfg, w, max_batch=8, max_text_len=25)
    mem = make_memory(5, cfg.mem_tokens, cfg.d_model, 31)
    ids = m.greedy_decode(mem, max_len=25, stop="never")
    assert torch.equal(ids, m.greedy_decode(mem, max_len=25, stop="never"))          # replayed graph
    full = m.forward_decoder(ids[:, :-1], mem)
    for b in (0, 3, 4):
        assert torch.equal(m.greedy_decode(mem[b:b + 1], max_len=25, stop="never"), ids[b:b + 1])
        assert torch.equal(m.forward_decoder(ids[b:b + 1, :-1], mem[b:b + 1]), full[b:b + 1])
    # a permutation of the batch permutes the result
    perm = torch.tensor([3, 0, 4, 1, 2])
    assert torch.equal(m.greedy_decode(mem[perm], max_len=25, stop="never"), ids[perm])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base"])
def test_gpu_student_row_prologue_switch_changes_nothing(name):
    """With one or two rows the row LayerNorms run inside the consuming projection (csrc/skinny.hip "row prologue"): switching
    that off at run time (the hipGraph is re-captured) must give the same ids bit for bit, alone and as the first rows of a
    batch of 4 (which never uses the prologue)."""
    from gitcap import _lib
    lib = _lib.load()
    cfg = student_tiny() if name == "tiny" else student_base()
    m = _student(cfg, student_synthetic_weights(cfg, 0), max_batch=4, max_text_len=16)
    mem = make_memory(4, cfg.mem_tokens, cfg.d_model, 9).cuda()
    full = m.greedy_decode(mem, max_len=16, stop="never")
    on = [m.greedy_decode(mem[:n], max_len=16, stop="never").clone() for n in (1, 2)]
    old = lib.gitcap_dbg_config(1, 0)
    try:
        off = [m.greedy_decode(mem[:n], max_len=16, stop="never").clone() for n in (1, 2)]
    finally:
        lib.gitcap_dbg_config(1, old)
    for a, b, n in zip(on, off, (1, 2)):
        assert torch.equal(a, b) and torch.equal(a, full[:n])
    # key 10: the vocabulary head as one single-wave workgroup per 16-column tile instead of four tiles sharing the rows in LDS
    old = lib.gitcap_dbg_config(10, 0)
    try:
        assert torch.equal(m.greedy_decode(mem, max_len=16, stop="never"), full)
    finally:
        lib.gitcap_dbg_config(10, old)
