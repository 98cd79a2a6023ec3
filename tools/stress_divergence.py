#!/usr/bin/env python
"""profiles/r06_stress_divergence.md: per stage, device vs bf16-emulating oracle with SHARED inputs (the oracle's stage is fed the
device's own stage input) next to the end-to-end distance -- shows that the end-to-end numbers of tests/test_stress_gpu.py
(GIT-base stress: 1.04 from the emulating oracle on logits of std 4) are amplification by the network, not a kernel's error.
Run on the GPU box:  python tools/stress_divergence.py gpurun_out/r06_stress_divergence.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-video-captioning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch                                                                     # noqa: E402
from stress_layers import device_stages, format_table, stage_table              # noqa: E402
from gitcap.config import git_base, git_large                                   # noqa: E402
from gitcap.model import GitCaptioner                                           # noqa: E402
from gitcap.weights import quantize_weights_fp8, stress_weights, synthetic_weights   # noqa: E402
from oracle.git_oracle import GitOracle, make_frames                            # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_stress_divergence.md"
    parts = ["# Stress-family divergence, stage by stage (round 6)\n",
             "Unit: bf16 ulps of the row maximum, ulp(row) = 2^(floor(log2 max|ref row|) - 7) (tests/stress_layers.py).  "
             "'shared inputs': the emulating oracle's stage is fed the DEVICE's input of that stage; 'end to end': the oracle's own "
             "chain.  Logit-level summary at the end of each table.\n"]
    cases = []
    cfg = git_base(2)
    g = torch.Generator().manual_seed(20)
    ids = torch.randint(1000, cfg.vocab_size, (2, 20), generator=g)
    ids[:, 0] = cfg.cls_token_id
    for fam, wf in (("plain", synthetic_weights), ("stress", stress_weights)):
        cases.append((f"GIT-base, {fam} weights, 2 clips x 2 frames, T = 20", cfg, wf(cfg, 0), make_frames(2, 2, cfg.image_size, 1234), ids, {}))
    cl = git_large(num_frames=10)
    cases.append(("GIT-large (BASELINE configs[4] shape), stress weights (e4m3-valued), 1 clip x 10 frames, T = 6", cl,
                  quantize_weights_fp8(stress_weights(cl, 0)), make_frames(1, 10, cl.image_size, 77),
                  torch.tensor([[101, 2023, 2003, 1037, 3899, 2006]]), {"weight_dtype": "fp8_e4m3"}))
    for title, cfg, w, fr, ids, kw in cases:
        m = GitCaptioner(cfg, w, device="cuda:0", max_batch=fr.shape[0], max_frames=fr.shape[1], max_text_len=24, **kw)
        dev = device_stages(m, fr, ids)
        emul, fp32 = GitOracle(cfg, w, emulate_bf16=True), GitOracle(cfg, w)
        rows = stage_table(cfg, dev, emul, fr, ids)
        with torch.no_grad():
            l_e, _ = emul.forward_output_logits(fr, ids)
            l_f, _ = fp32.forward_output_logits(fr, ids)
        lg = dev["logits"]
        txt = format_table(title, rows)
        txt += (f"\nLogits (std {float(l_f.std()):.2f}): end to end max |device - emul| {float((lg - l_e).abs().max()):.3f}, "
                f"|emul - fp32| {float((l_e - l_f).abs().max()):.3f}, |device - fp32| {float((lg - l_f).abs().max()):.3f}; "
                f"shared-input head stage max abs {rows[-1]['shared'][3]:.4g}.\n")
        print(txt, flush=True)
        parts.append(txt)
        del m
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
