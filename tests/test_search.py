"""Search operator: the oracle restatement of model.py:479-678 against exhaustive enumeration, and (GPU)
the product's GeneratorWithBeamSearch + beam_topk kernel against that oracle."""
import itertools
import numpy as np
import math
import os

import pytest
import torch

from oracle.search_oracle import beam_search as oracle_beam_search
from oracle.search_oracle import top_k_top_p_filtering
from oracle import search_oracle


def _toy_step(V, seed):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(8, V, V, generator=g) * 2.0          # logits depend on (position, last token)

    def step(ids):
        pos = ids.shape[1] - 1
        return table[pos % 8][ids[:, -1] % V].clone()
    return step


def _exhaustive_best(step, V, eos, cls, max_steps, length_penalty):
    """Best finished hypothesis by score = sum_logprob / len ** length_penalty over ALL sequences that
    end with EOS (or are cut at max length), as the reference scores them (hyp excludes the EOS)."""
    best = (-1e30, None)
    def rec(prefix, lp):
        nonlocal best
        cur_len = len(prefix)
        logp = torch.log_softmax(step(torch.tensor([prefix])).float(), -1)[0]
        for w in range(V):
            s = lp + float(logp[w])
            if w == eos or cur_len + 1 == max_steps:
                score = s / cur_len ** length_penalty
                if score > best[0]:
                    best = (score, list(prefix))
            else:
                rec(prefix + [w], s)
    rec([cls], 0.0)
    return best


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_beam_equals_exhaustive_when_beam_covers_vocab(seed):
    V, eos, cls, max_steps = 4, 3, 0, 5
    step = _toy_step(V, seed)
    # beam wide enough to keep every live prefix: 3 live words per node -> <= 27 prefixes at depth 3
    dec, lp, _ = oracle_beam_search(torch.tensor([[cls]]), step, eos_index=eos, max_steps=max_steps, beam_size=27,
                                    per_node_beam_size=V, length_penalty=0.6)
    score, seq = _exhaustive_best(step, V, eos, cls, max_steps, 0.6)
    assert math.isclose(float(lp[0, 0]), score, rel_tol=1e-5, abs_tol=1e-5)
    assert dec[0, :len(seq)].tolist() == seq and int(dec[0, len(seq)]) == eos


def test_oracle_beam_output_format_and_greedy_limit():
    V, eos, cls = 11, 10, 0
    step = _toy_step(V, 5)
    start = torch.tensor([[cls], [cls]])
    dec, lp, saved = oracle_beam_search(start, step, eos_index=eos, max_steps=7, beam_size=4, length_penalty=0.6)
    assert dec.shape == (2, 7) and lp.shape == (2, 1) and bool((dec[:, 0] == cls).all())
    assert torch.equal(dec[0], dec[1])                       # identical clips -> identical captions
    assert len(saved) >= 1 and saved[0].shape == (8, V)      # [B*beams, V] per step (model.py:521)
    # after the first EOS everything is EOS padding
    for row in dec.tolist():
        if eos in row:
            assert all(t == eos for t in row[row.index(eos):])


@pytest.mark.gpu
def test_beam_topk_kernel_vs_torch():
    import ctypes
    from gitcap import _lib
    lib = _lib.load()
    for B, beams, V, K in [(3, 4, 30522, 8), (2, 1, 197, 2), (1, 8, 1000, 16), (5, 3, 64, 6)]:
        g = torch.Generator(device="cuda").manual_seed(V)
        logits = torch.randn(B * beams, V, device="cuda", generator=g) * 3
        bs = torch.randn(B * beams, device="cuda", generator=g)
        out_s = torch.empty(B, K, device="cuda"); out_i = torch.empty(B, K, device="cuda", dtype=torch.int32)
        rc = lib.gitcap_beam_topk(ctypes.c_void_p(logits.data_ptr()), V, ctypes.c_void_p(bs.data_ptr()), B, beams, V, K,
                                  ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        ref = (torch.log_softmax(logits, -1) + bs[:, None]).view(B, beams * V)
        rs, ri = ref.topk(K, dim=1)
        assert torch.allclose(out_s, rs, atol=1e-4)
        assert torch.equal(out_i.long(), ri)
    # argument errors are reported, not launched
    assert lib.gitcap_beam_topk(None, 1, None, 1, 1, 1, 1, None, None, None) == -1


def test_oracle_top_k_top_p_filtering_known_answers():
    """Published HF algorithm behind model.py:537: hand-worked cases."""
    ninf = float("-inf")
    p = torch.tensor([[0.5, 0.25, 0.15, 0.07, 0.03]])
    lg = p.log()
    # top_k=1 but min_tokens_to_keep=2 keeps the two largest
    assert torch.equal(top_k_top_p_filtering(lg, top_k=1, min_tokens_to_keep=2) > ninf, torch.tensor([[True, True, False, False, False]]))
    # top_p=0.7: cumulative 0.5, 0.75 -> the token that crosses 0.7 is kept, the rest dropped
    assert torch.equal(top_k_top_p_filtering(lg, top_p=0.7) > ninf, torch.tensor([[True, True, False, False, False]]))
    # top_p=0.95 keeps 0.5+0.25+0.15+0.07 (0.97 crosses), drops the last
    assert torch.equal(top_k_top_p_filtering(lg, top_p=0.95) > ninf, torch.tensor([[True, True, True, True, False]]))
    # order of the input does not matter; kept values are unchanged
    perm = torch.tensor([3, 0, 4, 2, 1])
    out = top_k_top_p_filtering(lg[:, perm], top_k=3)
    assert torch.equal(out > ninf, torch.tensor([[False, True, False, True, True]])) and torch.equal(out[out > ninf], lg[:, perm][out > ninf])


def test_oracle_sampling_branch_degenerate_cases():
    """Sampling (model.py:532-554) with per_node_beam_size = 1 and a filter that leaves the two best tokens:
    every drawn token is one of the step's two largest logits; repetition penalty 1.0 is the identity."""
    V, eos, cls = 30, 29, 0
    step = _toy_step(V, 4)
    start = torch.tensor([[cls], [cls]])
    g = torch.Generator().manual_seed(1)
    dec, lp, _ = oracle_beam_search(start, step, eos_index=eos, max_steps=7, beam_size=2, per_node_beam_size=1,
                                    do_sample=True, top_k=2, generator=g)
    assert dec.shape == (2, 7) and bool((dec[:, 0] == cls).all())
    for row in dec.tolist():
        for t in range(1, len(row)):
            if row[t] == eos:
                break
            top2 = step(torch.tensor([row[:t]]))[0].topk(2).indices.tolist()
            assert row[t] in top2
    a = oracle_beam_search(start, step, eos_index=eos, max_steps=7, beam_size=3)
    b = oracle_beam_search(start, step, eos_index=eos, max_steps=7, beam_size=3, repetition_penalty=1.0, temperature=1.0)
    assert torch.equal(a[0], b[0])
    # the penalty rescales the logits of tokens already in the hypothesis (:522-531): the hypothesis scores change
    c = oracle_beam_search(start, step, eos_index=eos, max_steps=7, beam_size=3, repetition_penalty=5.0)
    assert c[0].shape == a[0].shape and not torch.allclose(c[1], a[1])


@pytest.mark.gpu
def test_product_sampling_and_repetition_penalty_vs_oracle():
    """Same CPU generator on both sides: the product's vectorised device filtering + host draw must reproduce the
    oracle's loop-style restatement of model.py:522-554 exactly on a toy step with well separated logits."""
    from gitcap.search import GeneratorWithBeamSearch
    V, eos, cls = 40, 39, 0
    step_cpu = _toy_step(V, 11)
    step_gpu = lambda ids: step_cpu(ids.cpu()).cuda()
    start = torch.tensor([[cls], [cls], [cls]])
    for kw in (dict(top_k=5), dict(top_p=0.8), dict(top_k=8, top_p=0.9)):
        for rp, temp in ((1.0, 1.0), (1.3, 0.7)):
            want = oracle_beam_search(start, step_cpu, eos_index=eos, max_steps=8, beam_size=3, length_penalty=0.6,
                                      repetition_penalty=rp, temperature=temp, do_sample=True,
                                      generator=torch.Generator().manual_seed(123), **kw)
            got = GeneratorWithBeamSearch(eos, 8, 3, length_penalty=0.6, repetition_penalty=rp, temperature=temp).search(
                start.cuda(), step_gpu, do_sample=True, generator=torch.Generator().manual_seed(123), **kw)
            assert torch.equal(got[0].cpu(), want[0]), (kw, rp, temp)
            assert torch.allclose(got[1].cpu(), want[1], atol=1e-4)
    # repetition penalty on the greedy-beam branch (HIP top-k kernel on the penalised scores)
    want = oracle_beam_search(start, step_cpu, eos_index=eos, max_steps=8, beam_size=4, length_penalty=0.6, repetition_penalty=1.5)
    got = GeneratorWithBeamSearch(eos, 8, 4, length_penalty=0.6, repetition_penalty=1.5).search(start.cuda(), step_gpu)
    assert torch.equal(got[0].cpu(), want[0]) and torch.allclose(got[1].cpu(), want[1], atol=1e-4)


@pytest.mark.gpu
def test_product_search_operator_vs_oracle_on_toy_step():
    from gitcap.search import GeneratorWithBeamSearch
    V, eos, cls = 50, 49, 0
    step_cpu = _toy_step(V, 9)
    step_gpu = lambda ids: step_cpu(ids.cpu()).cuda()
    start = torch.tensor([[cls], [cls], [cls]])
    want = oracle_beam_search(start, step_cpu, eos_index=eos, max_steps=9, beam_size=4, length_penalty=0.6)
    got = GeneratorWithBeamSearch(eos, 9, 4, length_penalty=0.6).search(start.cuda(), step_gpu, save_logits=True)
    assert torch.equal(got[0].cpu(), want[0])
    assert torch.allclose(got[1].cpu(), want[1], atol=1e-4)
    assert len(got[2]) == len(want[2])


@pytest.mark.gpu
def test_device_beam_search_vs_oracle_tiny_model():
    """End to end: encoder + KV-cached decoder steps + beam reorder on the device against the oracle's
    full-recompute step driven by the oracle search loop."""
    from gitcap.config import git_tiny
    from gitcap.model import GitCaptioner
    from gitcap.weights import synthetic_weights
    from oracle.git_oracle import GitOracle, make_frames
    cfg = git_tiny(2)
    w = synthetic_weights(cfg, 0)
    fr = make_frames(2, 2, cfg.image_size, 21)
    m = GitCaptioner(cfg, w, max_batch=2, max_text_len=10, max_beams=4)
    out = m.infer(fr, beam_size=4, max_steps=8, length_penalty=0.6, on_device=False)
    # device-resident search (gitcap_beam_search): same kernels produce the logits, so it must reproduce the
    # host-side operator exactly, for several shapes of the search
    for beams, steps, lp in [(4, 8, 0.6), (2, 6, 1.0), (1, 5, 0.6), (3, 10, 0.0)]:
        host = m.infer(fr, beam_size=beams, max_steps=steps, length_penalty=lp, on_device=False)
        dev = m.infer(fr, beam_size=beams, max_steps=steps, length_penalty=lp, on_device=True)
        assert torch.equal(dev["predictions"], host["predictions"]), (beams, steps, lp)
        assert torch.allclose(dev["logprobs"].cpu(), host["logprobs"].cpu(), atol=1e-5)
    orc = GitOracle(cfg, w, emulate_bf16=True)
    _, mem = orc.forward_image_enc(fr)

    def step(ids):                       # exact semantics: logits of the last position given the full prefix
        memb = mem.repeat_interleave(4, dim=0)
        return orc.decoder_full(memb, ids)[:, -1]
    want = oracle_beam_search(torch.full((2, 1), cfg.cls_token_id), step, eos_index=cfg.sep_token_id, max_steps=8,
                              beam_size=4, length_penalty=0.6)
    assert out["predictions"].shape == (2, 8)
    assert torch.allclose(out["logprobs"].cpu(), want[1], atol=0.05)          # bf16 operand noise on sum of log-probs
    if not torch.equal(out["predictions"].cpu(), want[0]):
        # a near-tie between hypotheses may legitimately pick another one: scores must then be within noise
        assert float((out["logprobs"].cpu() - want[1]).abs().max()) < 0.02
    assert torch.equal(m.beam_search(fr, max_len=8, k=4).cpu(), out["predictions"].cpu())
    # greedy still works on the same handle afterwards (slot state intact)
    assert m.greedy_decode(fr, max_len=6, stop="never").shape == (2, 7)


def _beam_golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "beam_tiny.npz"))
    cases = sorted(k[:-4] for k in g.files if k.endswith("_ids"))
    return g, cases


def test_oracle_reproduces_beam_golden():
    """SURVEY.md par. 8c fixture (3): the committed beam ids / log-probabilities of the GIT tiny config come from
    oracle/gen_golden_beam.py (restated search loop of model.py:479-678 over the fp32 oracle's step); the oracle
    must still reproduce them."""
    from oracle import gen_golden_beam as G
    g, cases = _beam_golden()
    assert len(cases) == 3
    with torch.no_grad():
        for c in G.CASES:
            dec, lp = G.run(c)
            assert np.array_equal(dec.numpy(), g[c["name"] + "_ids"]), c["name"]
            assert np.allclose(lp.numpy(), g[c["name"] + "_logprobs"], atol=1e-5)


@pytest.mark.gpu
def test_device_beam_search_vs_committed_golden():
    """gitcap_beam_search (device resident) and the host operator against the committed fp32 goldens: same
    hypothesis, or -- where bf16 operand noise flips a near-tie between hypotheses -- one that scores within noise."""
    from gitcap.config import git_tiny
    from gitcap.model import GitCaptioner
    from gitcap.weights import synthetic_weights
    from oracle.git_oracle import make_frames
    g, cases = _beam_golden()
    cfg = git_tiny(int(g["F"]))
    m = GitCaptioner(cfg, synthetic_weights(cfg, int(g["weight_seed"])), max_batch=2, max_text_len=16, max_beams=4)
    fr = make_frames(int(g["B"]), int(g["F"]), cfg.image_size, int(g["frame_seed"]))
    exact = 0
    for name in cases:
        beams, steps, lp = g[name + "_cfg"]
        for on_device in (True, False):
            out = m.infer(fr, beam_size=int(beams), max_steps=int(steps), length_penalty=float(lp), on_device=on_device)
            got, glp = out["predictions"].cpu().numpy(), out["logprobs"].cpu().numpy().reshape(-1)
            want, wlp = g[name + "_ids"], g[name + "_logprobs"].reshape(-1)
            assert got.shape == want.shape
            assert np.abs(glp - wlp).max() < 0.05, (name, glp, wlp)          # bf16 operand noise on a sum of log-probs
            for b in range(got.shape[0]):
                if np.array_equal(got[b], want[b]):
                    exact += 1
                else:
                    assert abs(glp[b] - wlp[b]) < 0.02, (name, b, got[b], want[b])
    assert exact >= 8, exact        # of 12 (3 cases x 2 clips x 2 drivers)


def test_oracle_teacher_output_known_answer():
    """oracle/search_oracle.py: teacher_output (model.py:771-788) on a hand-made case: per predicted word the logits of
    the beam that scores THAT word highest."""
    from oracle.search_oracle import teacher_output
    V, beams = 5, 4
    step0 = np.zeros((beams, V), np.float32); step0[2, 3] = 7.0; step0[2, 0] = -1.0      # word 3: beam 2 wins
    step1 = np.zeros((beams, V), np.float32); step1[0, 1] = 2.0; step1[3, 1] = 2.5      # word 1: beam 3 wins
    step2 = np.ones((beams, V), np.float32)                                             # never reached: cap has 2 words
    pred = torch.tensor([[101, 3, 1, 4, 102]])
    out = teacher_output(pred, [step0, step1, step2], "w3 w1")
    assert out.shape == (1, 2, V)
    assert np.array_equal(out[0, 0].numpy(), step0[2]) and np.array_equal(out[0, 1].numpy(), step1[3])
    # n is capped by the number of saved steps (model.py:772)
    assert teacher_output(pred, [step0], "w3 w1 w4").shape == (1, 1, V)


def test_cfg4_exact_fixture_certificate():
    """oracle/gen_golden_cfg4_beam.py: (1) the committed configs[4] fixture carries a certificate with a comfortable final
    margin; (2) the certificate means what it says -- on the tiny config, a certified caption survives every search whose
    candidate scores are perturbed by less than BAND / 2 (here: logits jittered so that 15 steps of log-softmax differences
    add up to less than that)."""
    import numpy as np
    from gitcap.config import git_tiny
    from gitcap.weights import synthetic_weights
    from oracle.git_oracle import GitOracle, make_frames
    from oracle.gen_golden_cfg4_beam import BAND, BEAMS, LENGTH_PENALTY, PER_NODE, STEPS, Reject, certify
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg4_beam_exact.npz"))
    assert float(g["final_margin"]) >= 0.05 and float(g["band"]) == BAND and g["predictions"].shape == (1, STEPS)
    assert int(g["predictions"][0, 0]) == 101 and int(g["beams"]) == BEAMS
    cfg = git_tiny(2)
    orc = GitOracle(cfg, synthetic_weights(cfg, 0), emulate_bf16=True)
    accepted = 0
    for seed in range(8):
        fr = make_frames(1, 2, cfg.image_size, seed)
        with torch.no_grad():
            _, mem = orc.forward_image_enc(fr)
            ikv = orc.image_kv(mem)

            def step(t):
                return orc.decoder_text(ikv, t, torch.zeros(t.shape[0], dtype=torch.long))[:, -1]
            try:
                c = certify(step, cfg)
            except Reject:
                continue
            accepted += 1
            gen = torch.Generator().manual_seed(seed)
            amp = BAND / 2 / (2 * STEPS) * 0.95                  # |d log-softmax| <= 2 amp per step, summed over <= STEPS steps
            for _ in range(12):
                def noisy(t):
                    l = step(t)
                    return l + (torch.rand(l.shape, generator=gen) * 2 - 1) * amp
                ids, _, _ = search_oracle.beam_search(torch.full((1, 1), cfg.cls_token_id), noisy, eos_index=cfg.sep_token_id,
                                                      max_steps=STEPS, beam_size=BEAMS, per_node_beam_size=PER_NODE,
                                                      length_penalty=LENGTH_PENALTY)
                assert ids[0, :len(c["ids"])].tolist() == list(c["ids"]), (seed, ids, c)
    assert accepted >= 1
