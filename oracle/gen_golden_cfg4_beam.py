"""Golden vector for BASELINE configs[4] that a device must reproduce EXACTLY.  TEST INFRASTRUCTURE ONLY.

The configs[4]-shape search case of rounds 2/3 (frame seed 41) parts from the oracle's search at a pruning decision whose
candidates tie within the bf16 score noise (docs/LAB_NOTEBOOK.md par. 3), so it can only be checked margin-gated.  This script looks,
on the CPU and with the oracle alone, for a clip whose CAPTION cannot depend on such ties: GIT-large (ViT-L/14; teacher
config /root/reference/data/teacher_configs/GIT_LARGE_MSRVTT/parameter.yaml:1-3), 10 frames, e4m3-valued weights (the very
weights of test_config4_real_shape_fp8_beam), beam 4, 15 steps, length_penalty 0.6 (search defaults
/root/reference/src/models/model.py:702-708), searched with the restated loop of model.py:479-678 over the bf16-emulating
oracle's text step.

Why not simply "every pruning margin > NEAR_TIE" (VERDICT r3 item 2): measured here (seeds 100-105), the 4th and 5th best of
the 4 x 30522 candidates of a step are 0.002-0.06 apart at SOME step of every search -- the low beams are near-permutations
of each other (one deviation from the best path, taken at different positions) and tie structurally.  Scaling the output head
scales the noise with the margins; planting peaked logits (round 4 tried: a +60 output-bias boost of six live words, the
decoder's residual writers damped 8x so that the previous word decides the next, position embeddings amplified 4x) sharpens
the best path and leaves the low beams -- and the winner's shifted copies -- tied (cut margins 0.004-0.07, final margins
0.002-0.08).  Those ties are harmless as long as no hypothesis that a differently-broken tie lets through can win.  That is
what is certified instead, on the UNMODIFIED configs[4] weights:

  ROBUSTNESS CERTIFICATE.  Let BAND = NEAR_TIE = 0.16 (twice the largest device-vs-oracle score difference measured at this
  shape, 0.08).  The search is re-run as a TREE: at every step a non-EOS candidate is "surely kept" when fewer than `beams`
  others reach to within BAND below it, "possibly kept" when fewer than `beams` others beat it by more than BAND; the tree
  branches over every way of filling the beam from the possibly-kept ones.  An EOS candidate (or, at the last step, any
  candidate: model.py:592) within BAND of the rank that decides whether it is examined is a "possible" finished hypothesis;
  the is_done test (model.py:575) branches when it is within BAND / max_length ** length_penalty of flipping.  A seed is
  accepted when, in EVERY leaf, the best surely-finished hypothesis is the same token sequence W, and no possible
  hypothesis with other tokens comes within FINAL_MARGIN = 0.05 (length-normalised; three times the normalised noise) of W.
  Any search whose candidate scores differ from the oracle's by less than BAND / 2 each follows one of the tree's paths,
  so it returns W.

Seeds 100-115 (round 4, 15-40 s each on 5 cores): 102, 110 and 112 certified (final margins 0.069 / 0.309 / 0.113, trees of
190 / 6 / 40 leaves), the others rejected (tree beyond 512 leaves).  The accepted seed with the largest final margin (110)
goes to tests/golden/cfg4_beam_exact.npz: seeds, W, the oracle's log-probability, the tree's size and margin.  tests/test_parity_gpu.py::test_config4_exact_fixture demands torch.equal ids
and |delta logprob| < 0.05 from the device-resident search.

    python oracle/gen_golden_cfg4_beam.py [--seeds 100:124] [--threads 6]
"""
from __future__ import annotations

import argparse
import itertools
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-video-captioning_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from gitcap.config import git_large                      # noqa: E402
from gitcap.weights import quantize_weights_fp8, synthetic_weights   # noqa: E402
from oracle import search_oracle                          # noqa: E402
from oracle.git_oracle import GitOracle, make_frames      # noqa: E402

FRAMES, BEAMS, PER_NODE, STEPS, LENGTH_PENALTY, WEIGHT_SEED = 10, 4, 2, 15, 0.6, 0
NEAR_TIE = 0.16
BAND = NEAR_TIE
FINAL_MARGIN = 0.05
MAX_LEAVES = 512


class Reject(Exception):
    pass


def certify(step_fn, cfg):
    """Tree search described in the module docstring, for one clip.  Returns a dict (winner ids, its normalised score,
    smallest final margin, leaves) or raises Reject."""
    eos, V = cfg.sep_token_id, cfg.vocab_size
    max_len = STEPS
    norm_len = (max_len - 1) ** LENGTH_PENALTY
    leaves = []          # (winner_score, winner_ids) per leaf
    possible = []        # every hypothesis any leaf may finish: (ids, normalised score)
    memo_logp = {}

    def logp_of(prefixes):
        key = tuple(prefixes)
        if key not in memo_logp:
            memo_logp[key] = Fn.log_softmax(step_fn(torch.tensor(prefixes, dtype=torch.long)).float(), -1)
        return memo_logp[key]

    def walk(prefixes, scores, best, cur_len):
        """prefixes: BEAMS tuples; scores: their cumulative log-probs; best: (score, ids) of the surely-finished hypothesis
        the n_hyp = 1 container holds, or None."""
        if len(leaves) > MAX_LEAVES:
            raise Reject("tree too large")
        if cur_len >= max_len:
            leaves.append(best)
            return
        lp = logp_of(prefixes)
        cand = (lp + torch.tensor(scores)[:, None]).flatten()
        top = cand.topk(4 * BEAMS)
        tv, ti = top.values.tolist(), top.indices.tolist()
        # model.py:575: done = done or is_done(best candidate score)
        if best is not None:
            d = best[0] - tv[0] / norm_len
            if abs(d) <= BAND / norm_len:
                leaves.append(best)                         # the "done" branch ends here; the other continues below
            elif d >= 0:
                leaves.append(best)
                return
        last = cur_len + 1 == max_len
        hyp_len = cur_len ** LENGTH_PENALTY
        if last:                                            # every examined candidate finishes its prefix (model.py:592)
            # examined = the PER_NODE * BEAMS best; the winner among them is the best one, the rest can only matter as rivals
            fin = [(tv[k] / hyp_len, prefixes[ti[k] // V]) for k in range(len(tv))]
            nb = best
            if nb is None or fin[0][0] > nb[0]:
                nb = fin[0]
            for sc, ids in fin:
                possible.append((ids, sc))
            leaves.append(nb)
            return
        non_eos = [(s, i) for s, i in zip(tv, ti) if i % V != eos]
        eos_c = [(s, i) for s, i in zip(tv, ti) if i % V == eos]
        if len(non_eos) < BEAMS + 1:
            raise Reject("not enough candidates looked at")
        sure, maybe = [], []
        for k, (s, i) in enumerate(non_eos):
            others = [x for j, (x, _) in enumerate(non_eos) if j != k]
            if sum(1 for x in others if x >= s - BAND) < BEAMS:
                sure.append((s, i))
            elif sum(1 for x in others if x > s + BAND) < BEAMS:
                maybe.append((s, i))
        if non_eos[-1] in maybe or non_eos[-1] in sure:
            raise Reject("candidate window too small for the band")
        need = BEAMS - len(sure)
        for pick in itertools.combinations(maybe, need):
            kept = sorted(sure + list(pick), reverse=True)
            cut = kept[-1][0]
            nb = best
            for s, i in eos_c:                              # finished hypotheses: the prefix, scored with the EOS candidate
                sc, ids = s / hyp_len, prefixes[i // V]
                if s > cut + BAND:                          # surely examined before the beam is full
                    possible.append((ids, sc))
                    if nb is None or sc > nb[0]:
                        nb = (sc, ids)
                elif s >= cut - BAND:                       # examined or not, depending on the tie
                    possible.append((ids, sc))
            nprefixes = [prefixes[i // V] + (i % V,) for _, i in kept]
            walk(nprefixes, [s for s, _ in kept], nb, cur_len + 1)

    start = (cfg.cls_token_id,)
    walk([start] * BEAMS, [0.0] + [-1e9] * (BEAMS - 1), None, 1)
    if any(l is None for l in leaves):
        raise Reject("a leaf without a finished hypothesis")
    W = max(leaves, key=lambda l: l[0])
    for sc, ids in leaves:
        if ids != W[1]:
            raise Reject(f"leaves disagree ({len(leaves)} leaves)")
    w_score = min(sc for sc, _ in leaves)
    rivals = [sc for ids, sc in possible if ids != W[1]]
    margin = w_score - max(rivals) if rivals else 9.9
    if margin < FINAL_MARGIN:
        raise Reject(f"final margin {margin:.3f}, {len(leaves)} leaves")
    return {"ids": W[1], "score": W[0], "margin": margin, "leaves": len(leaves), "steps_evaluated": len(memo_logp)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="100:124")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "cfg4_beam_exact.npz"))
    args = ap.parse_args()
    lo, hi = (int(x) for x in args.seeds.split(":"))
    torch.set_num_threads(args.threads)
    cfg = git_large(num_frames=FRAMES)
    emul = GitOracle(cfg, quantize_weights_fp8(synthetic_weights(cfg, WEIGHT_SEED)), emulate_bf16=True)
    best = None
    for seed in range(lo, hi):
        t0 = time.time()
        fr = make_frames(1, FRAMES, cfg.image_size, seed)
        with torch.no_grad():
            _, mem = emul.forward_image_enc(fr)
            ikv = emul.image_kv(mem)

            def step(t):
                return emul.decoder_text(ikv, t, torch.zeros(t.shape[0], dtype=torch.long))[:, -1]
            try:
                c = certify(step, cfg)
            except Reject as e:
                print(f"seed {seed}: rejected ({e}) ({time.time() - t0:.0f} s)", flush=True)
                continue
            # the plain search must agree with the certificate (it is one path of the tree)
            ids, logprob, _ = search_oracle.beam_search(torch.full((1, 1), cfg.cls_token_id), step, eos_index=cfg.sep_token_id,
                                                        max_steps=STEPS, beam_size=BEAMS, per_node_beam_size=PER_NODE,
                                                        length_penalty=LENGTH_PENALTY)
        n = len(c["ids"])
        assert ids[0, :n].tolist() == list(c["ids"]) and abs(float(logprob[0, 0]) - c["score"]) < 1e-4, (ids, c)
        print(f"seed {seed}: ACCEPT ids {ids[0].tolist()} logprob {float(logprob[0, 0]):.4f} final margin {c['margin']:.3f} "
              f"leaves {c['leaves']} step evaluations {c['steps_evaluated']} ({time.time() - t0:.0f} s)", flush=True)
        if best is None or c["margin"] > best[1]["margin"]:
            best = (seed, c, ids, float(logprob[0, 0]))
    if best is None:
        raise SystemExit("no seed in the range could be certified; widen --seeds")
    seed, c, ids, logprob = best
    np.savez_compressed(args.out, frame_seed=np.int64(seed), weight_seed=np.int64(WEIGHT_SEED), frames=np.int64(FRAMES),
                        beams=np.int64(BEAMS), per_node_beam_size=np.int64(PER_NODE), max_steps=np.int64(STEPS),
                        length_penalty=np.float64(LENGTH_PENALTY), predictions=ids.numpy(), logprob=np.float64(logprob),
                        band=np.float64(BAND), final_margin=np.float64(c["margin"]), tree_leaves=np.int64(c["leaves"]),
                        step_evaluations=np.int64(c["steps_evaluated"]))
    print("wrote", args.out, "seed", seed)


if __name__ == "__main__":
    main()
