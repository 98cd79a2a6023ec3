"""Generates tests/golden/beam_tiny.npz: SURVEY.md par. 8c fixture (3) -- beam = 4 ids / log-probabilities of the GIT
tiny config from the restated search loop (oracle/search_oracle.py = /root/reference/src/models/model.py:479-678)
driven by the fp32 oracle's full-recompute step (oracle/git_oracle.py; itself pinned to the transformers goldens).
TEST INFRASTRUCTURE ONLY.  Weights and frames are regenerated from seeds, only ids and scores are stored.

    python oracle/gen_golden_beam.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-video-captioning_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from gitcap.config import git_tiny                    # noqa: E402
from gitcap.weights import synthetic_weights          # noqa: E402
from oracle.git_oracle import GitOracle, make_frames  # noqa: E402
from oracle.search_oracle import beam_search as oracle_beam_search   # noqa: E402

CASES = [dict(name="b4_s8", beams=4, steps=8, lp=0.6), dict(name="b4_s15", beams=4, steps=15, lp=0.6),
         dict(name="b2_s6", beams=2, steps=6, lp=1.0)]
WEIGHT_SEED, FRAME_SEED, B, F = 0, 21, 2, 2


def run(case, emulate_bf16=False):
    cfg = git_tiny(F)
    w = synthetic_weights(cfg, WEIGHT_SEED)
    fr = make_frames(B, F, cfg.image_size, FRAME_SEED)
    orc = GitOracle(cfg, w, emulate_bf16=emulate_bf16)
    _, mem = orc.forward_image_enc(fr)

    def step(ids):
        return orc.decoder_full(mem.repeat_interleave(case["beams"], dim=0), ids)[:, -1]
    dec, lp, saved = oracle_beam_search(torch.full((B, 1), cfg.cls_token_id), step, eos_index=cfg.sep_token_id,
                                        max_steps=case["steps"], beam_size=case["beams"], length_penalty=case["lp"])
    return dec, lp


if __name__ == "__main__":
    out = {"weight_seed": WEIGHT_SEED, "frame_seed": FRAME_SEED, "B": B, "F": F}
    with torch.no_grad():
        for c in CASES:
            dec, lp = run(c)
            out[c["name"] + "_ids"] = dec.numpy().astype(np.int64)
            out[c["name"] + "_logprobs"] = lp.numpy().astype(np.float32)
            out[c["name"] + "_cfg"] = np.array([c["beams"], c["steps"], c["lp"]], np.float32)
            print(c["name"], dec.tolist(), lp.flatten().tolist())
    np.savez(os.path.join(ROOT, "tests", "golden", "beam_tiny.npz"), **out)
