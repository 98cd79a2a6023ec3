"""Writes tests/golden/student_*.npz: outputs of ``torch.nn.TransformerDecoder`` -- the module the
reference's student instantiates (/root/reference/src/models/model.py:82-85) -- called the way
model.py:128-154 calls it, with the reference's own mask helpers imported from
/root/reference/src/utils/masking.py.  TEST INFRASTRUCTURE: run here (the reference tree does not exist
on the GPU box); only the vectors are committed.

    python oracle/gen_golden_student.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "real-time-video-captioning_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from src.utils.masking import create_casual_mask, create_padding_mask   # noqa: E402  (the reference's helpers)
from gitcap.student_config import student_base, student_stress_weights, student_synthetic_weights, student_tiny   # noqa: E402
from oracle.student_oracle import make_memory   # noqa: E402


class TorchStudentDecoder(nn.Module):
    """nn modules with the reference's attribute names, so its state_dict keys are the reference's."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.decoder_layer = nn.TransformerDecoderLayer(d_model=cfg.d_model, nhead=cfg.n_head, dim_feedforward=cfg.d_ffn,
                                                        dropout=0.3, batch_first=True)
        self.decoder = nn.TransformerDecoder(self.decoder_layer, cfg.num_decoder_layers)
        self.embed = nn.Embedding(cfg.vocab_length, cfg.d_model)
        self.linear = nn.Linear(cfg.d_model, cfg.vocab_length)
        self.register_buffer("pe", torch.zeros(1, cfg.max_pos, cfg.d_model))

    def load(self, weights):
        sd = {k: torch.from_numpy(v) for k, v in weights.items() if k != "pos_enc.pe"}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("decoder_layer.") or k == "pe" for k in missing), (missing, unexpected)
        self.pe.copy_(torch.from_numpy(weights["pos_enc.pe"]))

    def forward_decoder(self, y, memory):
        pad_mask = create_padding_mask(y)
        tgt_mask = create_casual_mask(y.shape[1])
        tgt = self.embed(y) + self.pe[:, : y.size(1)]
        tgt = tgt / torch.sqrt(torch.tensor(self.embed.embedding_dim))
        out = self.decoder(tgt=tgt, memory=memory, tgt_mask=tgt_mask, tgt_key_padding_mask=pad_mask, tgt_is_causal=True)
        return self.linear(out)

    def greedy(self, memory, max_len, stop_all_sep=True):
        tgt = torch.full((memory.shape[0], 1), self.cfg.cls_token_id, dtype=torch.long)
        for _ in range(max_len):
            last = self.forward_decoder(tgt, memory).argmax(-1)[:, -1:]
            tgt = torch.cat([tgt, last], dim=1)
            if stop_all_sep and bool((last.squeeze(-1) == self.cfg.sep_token_id).all()):
                break
        return tgt


    def beam_search(self, memory, max_len, k):
        """Loop-for-loop restatement of model.py:189-318 (all_candidates table, per-(b, idx) copy) -- kept
        deliberately unlike the oracle's vectorised form so the two check each other."""
        B = memory.shape[0]
        tgt = torch.full((B, 1), self.cfg.cls_token_id, dtype=torch.long)
        sequences = tgt.unsqueeze(1).expand(-1, k, -1)
        all_candidates = torch.empty(B, k * k, 3)
        log_probs = torch.log_softmax(self.forward_decoder(tgt, memory)[:, -1, :], dim=-1)
        scores, top_indices = log_probs.topk(k, dim=-1)
        sequences = torch.cat([sequences, top_indices.unsqueeze(-1)], dim=-1)
        for step in range(2, max_len):
            for i in range(k):
                log_probs = torch.log_softmax(self.forward_decoder(sequences[:, i], memory)[:, -1, :], dim=-1)
                top_scores, top_indices = log_probs.topk(k, dim=-1)
                all_candidates[:, i * k:(i + 1) * k, 0] = scores[:, i].unsqueeze(-1) + top_scores
                all_candidates[:, i * k:(i + 1) * k, 1] = i
                all_candidates[:, i * k:(i + 1) * k, 2] = top_indices
            order = all_candidates[:, :, 0].sort(dim=1, descending=True).indices[:, :k]
            new_sequences = torch.zeros(B, k, step + 1, dtype=torch.long)
            for b in range(B):
                for idx in range(k):
                    g = order[b, idx]
                    new_sequences[b, idx, :-1] = sequences[b, all_candidates[b, g, 1].long(), :]
                    new_sequences[b, idx, -1] = all_candidates[b, g, 2].long()
                    scores[b, idx] = all_candidates[b, g, 0]
            sequences = new_sequences
        return sequences[torch.arange(B), scores.argmax(dim=-1)]


@torch.no_grad()
def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    # ---- tiny: full logits, PAD tokens in the teacher-forced prefix, greedy ids ----------------------
    cfg = student_tiny()
    w = student_synthetic_weights(cfg, 0)
    m = TorchStudentDecoder(cfg).eval()
    m.load(w)
    mem = make_memory(3, cfg.mem_tokens, cfg.d_model, 11)
    y = torch.tensor([[1, 5, 9, 33, 7, 2], [1, 77, 0, 15, 0, 4], [1, 3, 3, 0, 0, 0]])
    np.savez(os.path.join(out_dir, "student_tiny.npz"), mem_seed=11, y=y.numpy(), logits=m.forward_decoder(y, mem).numpy(),
             greedy_ids=m.greedy(mem, 12, stop_all_sep=False).numpy(), beam_k3=m.beam_search(mem, 9, 3).numpy(),
             beam_k4=m.beam_search(mem, 6, 4).numpy())
    # a head bias that makes PAD (0) the arg-max at every step: later steps see all earlier keys but CLS masked
    w2 = dict(w); w2["linear.bias"] = w["linear.bias"].copy(); w2["linear.bias"][cfg.pad_token_id] = 50.0
    m.load(w2)
    y2 = torch.tensor([[1, 0, 0, 0, 0]])
    np.savez(os.path.join(out_dir, "student_tiny_pad.npz"), mem_seed=11, y=y2.numpy(),
             logits=m.forward_decoder(y2, mem[:1]).numpy(), greedy_ids=m.greedy(mem[:1], 6, stop_all_sep=False).numpy())
    # ---- base (config.py:78-83): seeds + slices -------------------------------------------------------
    cfg = student_base()
    w = student_synthetic_weights(cfg, 0)
    m = TorchStudentDecoder(cfg).eval()
    m.load(w)
    mem = make_memory(2, cfg.mem_tokens, cfg.d_model, 12)
    ids = m.greedy(mem, 25, stop_all_sep=False)
    logits = m.forward_decoder(ids[:, :-1], mem)
    top_v, top_i = logits.topk(8, dim=-1)
    np.savez(os.path.join(out_dir, "student_base.npz"), mem_seed=12, greedy_ids=ids.numpy(), top_ids=top_i.numpy(),
             top_vals=top_v.numpy(), first16=logits[:, :, :16].numpy())
    # ---- the stress family (student_stress_weights: outlier LayerNorm channels, big ReLU inputs, a peaked head), both sizes ----
    for name, cfg, B, L, seed in (("tiny", student_tiny(), 3, 12, 13), ("base", student_base(), 2, 25, 14)):
        w = student_stress_weights(cfg, 0)
        m = TorchStudentDecoder(cfg).eval()
        m.load(w)
        mem = make_memory(B, cfg.mem_tokens, cfg.d_model, seed)
        ids = m.greedy(mem, L, stop_all_sep=False)
        logits = m.forward_decoder(ids[:, :-1], mem)
        top_v, top_i = logits.topk(8, dim=-1)
        np.savez(os.path.join(out_dir, f"student_{name}_stress.npz"), mem_seed=seed, greedy_ids=ids.numpy(), top_ids=top_i.numpy(),
                 top_vals=top_v.numpy(), first16=logits[:, :, :16].numpy())
    print("wrote student goldens to", out_dir)


if __name__ == "__main__":
    main()
