"""What would fp8 MFMA compute cost in accuracy?  TEST INFRASTRUCTURE ONLY (a CPU study with the oracle; nothing ships).

VERDICT r3 item 3 asked for an opt-in fp8 image pass for BASELINE configs[4] (e4m3 weights read as bytes, activations
quantised to e4m3 with a per-row power-of-two scale, `v_mfma_scale_f32_*_f8f6f4`), with a stop rule: "if activation rounding
pushes |dlogit| past 0.3 (7 % of the spread) vs the bf16-emulating oracle, record that and stop".  This script measures that
number BEFORE any kernel is written: the oracle with `emulate_fp8_act=True` rounds the activation operand of the chosen
image-row GEMMs to e4m3 (round to nearest even, per-row power-of-two scale = the best e4m3 can do short of per-element
scales; the text rows and the patch embedding stay bf16) and is compared with the bf16-emulating oracle on teacher-forced
logits of GIT-large (e4m3-valued weights, the configs[4] model) and GIT-base.

    python oracle/fp8_act_study.py > profiles/r04_fp8_activation_study.txt
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-video-captioning_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from gitcap.config import git_base, git_large                        # noqa: E402
from gitcap.weights import quantize_weights_fp8, synthetic_weights   # noqa: E402
import oracle.git_oracle as go                                       # noqa: E402
from oracle.git_oracle import GitOracle, make_frames                 # noqa: E402


class Subset(GitOracle):
    """e4m3 activations only in the image-row GEMMs whose weight name `pick` accepts."""

    def __init__(self, *a, pick=None, **k):
        super().__init__(*a, **k)
        self.pick = pick

    def _lin(self, x, name, img=False):
        use = img and self.f8 and (self.pick is None or self.pick(name))
        xin = go._q8(go._r(x, self.bf)) if use else go._r(x, self.bf)
        return torch.nn.functional.linear(xin, self.w[name + ".w"], self.w[name + ".b"])


PICKS = {
    "every image-row GEMM": None,
    "ViT encoder only": lambda n: n.startswith("enc."),
    "decoder image rows + projection only": lambda n: n.startswith("dec.") or n == "vproj",
    "LayerNorm-fed GEMMs only (qkv, fc1, vproj)": lambda n: n.endswith("qkv") or n.endswith("fc1") or n == "vproj",
    "context / GELU-fed GEMMs only (proj, ao, fc2)": lambda n: n.endswith("proj") or n.endswith("ao") or n.endswith("fc2"),
    "fc1 + fc2 only (2/3 of the GEMM FLOPs)": lambda n: n.endswith("fc1") or n.endswith("fc2"),
}


def main():
    torch.set_num_threads(int(os.environ.get("THREADS", "6")))
    ids = torch.tensor([[101, 2023, 2003, 1037, 3899], [101, 1037, 2158, 2006, 1996]])
    for name, cfg, F, seeds in (("GIT-base", git_base(2), 2, (41, 42, 43)), ("GIT-large", git_large(3), 3, (41, 42, 43))):
        w = quantize_weights_fp8(synthetic_weights(cfg, 0))
        res = {k: [] for k in PICKS}
        bf_vs_fp32, spread = [], []
        for seed in seeds:
            fr = make_frames(2, F, cfg.image_size, seed)

            def run(o):
                with torch.no_grad():
                    _, mem = o.forward_image_enc(fr)
                    return o.decoder_text(o.image_kv(mem), ids)
            full = run(GitOracle(cfg, w))
            base = run(GitOracle(cfg, w, emulate_bf16=True))
            bf_vs_fp32.append(float((base - full).abs().max()))
            spread.append(float(full.std()))
            for k, pk in PICKS.items():
                d = (run(Subset(cfg, w, emulate_bf16=True, emulate_fp8_act=True, pick=pk)) - base).abs()
                res[k].append((float(d.max()), float(d.mean())))
        print(f"{name}, {F} frames x 2 clips, 5 teacher-forced positions, e4m3-valued weights, frame seeds {seeds}: logit std "
              f"{sum(spread) / len(spread):.2f}; bf16-emulating oracle vs fp32 oracle max |dlogit| {max(bf_vs_fp32):.3f}")
        for k, v in res.items():
            print(f"  e4m3 activations in {k:48s}: max |dlogit| vs the bf16-emulating oracle {max(x[0] for x in v):.3f} "
                  f"(per seed {[round(x[0], 3) for x in v]}), mean {sum(x[1] for x in v) / len(v):.4f}", flush=True)


if __name__ == "__main__":
    main()
