"""Generate tests/golden/*.npz by running the `transformers` implementation of GIT
(transformers.models.git, HF port of microsoft/GenerativeImage2Text) on OUR seeded synthetic
weights and inputs.  TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/gen_golden_hf.py            # tiny + base + stress fixtures (~8 min of CPU); or: tiny | base | stress

The reference tree has no fixture for this path and its arithmetic package is
absent, so these vectors pin the oracle (oracle/git_oracle.py) to the one independent
implementation available offline.  Only `use_cache=False` full-recompute forwards are used
(transformers 5.15 offsets text position ids in its cached path, modeling_git.py:793-794).
Weights/inputs are NOT stored: they are regenerated from seeds (numpy PCG64) by
gitcap.weights.synthetic_weights / oracle.git_oracle.make_frames.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "real-time-video-captioning_amd"))
sys.path.insert(0, ROOT)

from gitcap.config import git_base, git_tiny, GitCapConfig          # noqa: E402
from gitcap.weights import stress_weights, synthetic_weights, to_hf_state_dict      # noqa: E402
from oracle.git_oracle import make_frames                            # noqa: E402


def build_hf(cfg: GitCapConfig, w):
    from transformers.models.git.configuration_git import GitConfig, GitVisionConfig
    from transformers.models.git.modeling_git import GitForCausalLM
    vc = GitVisionConfig(hidden_size=cfg.enc_width, intermediate_size=cfg.enc_ffn,
                         num_hidden_layers=cfg.enc_layers, num_attention_heads=cfg.enc_heads,
                         image_size=cfg.image_size, patch_size=cfg.patch_size,
                         hidden_act="quick_gelu", layer_norm_eps=cfg.enc_ln_eps)
    hc = GitConfig(vision_config=vc.to_dict(), vocab_size=cfg.vocab_size, hidden_size=cfg.dec_width,
                   num_hidden_layers=cfg.dec_layers, num_attention_heads=cfg.dec_heads,
                   intermediate_size=cfg.dec_ffn, hidden_act="gelu",
                   max_position_embeddings=cfg.max_text_pos, layer_norm_eps=cfg.dec_ln_eps,
                   num_image_with_embedding=(cfg.num_frames or None),
                   bos_token_id=cfg.cls_token_id, eos_token_id=cfg.sep_token_id,
                   pad_token_id=cfg.pad_token_id, tie_word_embeddings=False)
    hc._attn_implementation = "eager"
    model = GitForCausalLM(hc).eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in to_hf_state_dict(cfg, w).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    return model


@torch.no_grad()
def hf_logits(model, frames, ids):
    pv = frames if frames.shape[1] > 1 or model.config.num_image_with_embedding else frames[:, 0]
    if model.config.num_image_with_embedding is None:
        pv = frames[:, 0]
    out = model(input_ids=ids, pixel_values=pv, use_cache=False)
    T = ids.shape[1]
    return out.logits[:, -T:, :].float()


@torch.no_grad()
def hf_visual(model, frames):
    """ln_post output + temporal embedding, concatenated along tokens (modeling_git.py:805-824)."""
    git = model.git
    if git.config.num_image_with_embedding is None:
        return git.image_encoder(frames[:, 0]).last_hidden_state
    feats = []
    for f in range(frames.shape[1]):
        v = git.image_encoder(frames[:, f]).last_hidden_state
        feats.append(v + git.img_temporal_embedding[f])
    return torch.cat(feats, dim=1)


@torch.no_grad()
def hf_greedy(model, frames, max_len, cls_id):
    B = frames.shape[0]
    ids = torch.full((B, 1), cls_id, dtype=torch.long)
    last, top_i, top_v = [], [], []
    for _ in range(max_len):
        lg = hf_logits(model, frames, ids)[:, -1]
        v, i = lg.topk(8, dim=-1)
        last.append(lg[:, :16].clone()); top_i.append(i); top_v.append(v)
        ids = torch.cat([ids, lg.argmax(-1, keepdim=True)], 1)
    return ids, torch.stack(last, 1), torch.stack(top_i, 1), torch.stack(top_v, 1)


def gen_tiny(out_dir, stress=False):
    """stress=True: the second weight family (gitcap.weights.stress_weights: outlier LayerNorm channels, saturating GELU
    inputs, large-norm CLS / position rows, a peaked head) -> hf_tiny_stress.npz (F = 2 only)."""
    for F in ((2,) if stress else (2, 0)):
        cfg = git_tiny(num_frames=F)
        w = (stress_weights if stress else synthetic_weights)(cfg, seed=0)
        model = build_hf(cfg, w)
        Fr = max(1, F)
        frames = make_frames(2, Fr, cfg.image_size, seed=1234)
        g = np.random.default_rng(7)
        ids = torch.from_numpy(g.integers(1, cfg.vocab_size, size=(2, 6))).long()
        ids[:, 0] = cfg.cls_token_id
        logits = hf_logits(model, frames, ids)
        visual = hf_visual(model, frames)
        proj = model.git.visual_projection(visual)
        gids, last16, ti, tv = hf_greedy(model, frames, 8, cfg.cls_token_id)
        # per-layer hidden states over [image ; text] (what forward_one_custom stacks, model.py:419-424): the
        # encoder's input followed by the output of each of its layers, L+1 entries of [B, S_img + T, D]
        pv = frames if model.config.num_image_with_embedding is not None else frames[:, 0]
        hs = model(input_ids=ids, pixel_values=pv, use_cache=False, output_hidden_states=True).hidden_states
        hidden = torch.stack([h.float() for h in hs], 0)
        np.savez_compressed(os.path.join(out_dir, "hf_tiny_stress.npz" if stress else f"hf_tiny_F{F}.npz"), hidden=hidden.detach().numpy(),
                            prefix_ids=ids.numpy(), logits=logits.numpy(), visual=visual.numpy(),
                            projected=proj.detach().numpy(), greedy_ids=gids.numpy(),
                            greedy_top_ids=ti.numpy(), greedy_top_vals=tv.numpy(),
                            weight_seed=0, frame_seed=1234)
        print(f"tiny F={F}{' stress' if stress else ''}: logits {tuple(logits.shape)} greedy {gids.tolist()}")


def gen_base(out_dir, stress=False):
    """stress=True: GIT-base, 2 clips x 2 frames on gitcap.weights.stress_weights -> hf_base_F2_stress.npz."""
    for F, fname in (((2, "hf_base_F2_stress.npz"),) if stress else ((0, "hf_base_F1.npz"), (6, "hf_base_F6.npz"))):
        t0 = time.time()
        cfg = git_base(num_frames=F)
        w = (stress_weights if stress else synthetic_weights)(cfg, seed=0)
        model = build_hf(cfg, w)
        del w
        frames = make_frames(2, max(1, F), cfg.image_size, seed=1234)
        gids, last16, ti, tv = hf_greedy(model, frames, 20, cfg.cls_token_id)
        visual = hf_visual(model, frames)
        np.savez_compressed(os.path.join(out_dir, fname),
                            greedy_ids=gids.numpy(), greedy_last16=last16.numpy(),
                            greedy_top_ids=ti.numpy(), greedy_top_vals=tv.numpy(),
                            visual_slice=visual[:, ::97, :32].numpy(),
                            weight_seed=0, frame_seed=1234)
        print(f"base F={F}: greedy {gids.tolist()}  ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "tiny"):
        gen_tiny(out)
    if which in ("all", "base"):
        gen_base(out)
    if which in ("all", "stress"):
        gen_tiny(out, stress=True)
        gen_base(out, stress=True)
