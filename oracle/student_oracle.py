"""CPU oracle for the STUDENT caption decoder (SURVEY.md par. 8 row f.2).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing in the product path).

Restates, with plain tensor arithmetic, what the reference computes through ``torch.nn`` modules:
  * ``StudentCandidateV1.forward_decoder``  /root/reference/src/models/model.py:128-154
      embed -> + positional table (model.py:320-340) -> divide by sqrt(d_model) (order as written:
      the positional term is divided too) -> nn.TransformerDecoder (post-LN, ReLU, no final norm;
      model.py:82-85) with a causal mask and PAD tokens masked as keys
      (/root/reference/src/utils/masking.py:4-27) -> Linear to the vocabulary;
  * ``StudentCandidateV1.greedy_decode``  model.py:156-187
      CLS start, full recompute per step, argmax of the last position, stop iff every row emitted SEP
      in the same step; rows keep generating after their own SEP.

Pinned (tests/test_student.py) against ``torch.nn.TransformerDecoder`` itself -- the module the
reference instantiates -- called the way model.py:149-150 calls it with the reference's own mask
helpers, through the fixtures written by oracle/gen_golden_student.py.

``emulate_bf16=True`` rounds to bf16 exactly where the HIP path does (GEMM operands: weights,
layer inputs, q/k/v, attention context, FFN hidden, memory), everything else fp32.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class StudentOracle:
    def __init__(self, cfg, weights: Dict[str, np.ndarray], emulate_bf16: bool = False):
        self.cfg = cfg
        self.emu = emulate_bf16
        gemm = ("in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight")
        self.w = {}
        for k, v in weights.items():
            t = torch.from_numpy(np.ascontiguousarray(v)).float()
            if emulate_bf16 and (k.endswith(gemm) or k == "linear.weight"):
                t = _bf(t)
            self.w[k] = t

    def _r(self, x):
        return _bf(x) if self.emu else x

    def _ln(self, x, p):
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), self.w[p + ".weight"], self.w[p + ".bias"], self.cfg.ln_eps)

    def _heads(self, x):                       # [B, T, D] -> [B, H, T, hd]
        B, T, _ = x.shape
        return x.view(B, T, self.cfg.n_head, self.cfg.head_dim).transpose(1, 2)

    def _attend(self, q, k, v, mask):          # mask: additive [B, 1, Tq, Tk] or None
        s = torch.matmul(self._heads(q), self._heads(k).transpose(-1, -2)) / math.sqrt(self.cfg.head_dim)
        if mask is not None:
            s = s + mask
        p = torch.softmax(s, dim=-1)
        ctx = torch.matmul(p, self._heads(v)).transpose(1, 2)
        return self._r(ctx.reshape(q.shape[0], q.shape[1], self.cfg.d_model))

    def forward_decoder(self, y: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
        """y [B, T] int64, memory [B, F, D] -> logits [B, T, V]  (model.py:128-154)."""
        c, w, D = self.cfg, self.w, self.cfg.d_model
        B, T = y.shape
        pad = (y == c.pad_token_id)                                            # masking.py:14
        causal = torch.triu(torch.ones(T, T), diagonal=1).bool()               # masking.py:27
        mask = torch.zeros(B, 1, T, T)
        mask = mask.masked_fill(causal[None, None], float("-inf")).masked_fill(pad[:, None, None, :], float("-inf"))
        x = w["embed.weight"][y] + w["pos_enc.pe"][0, :T]                      # model.py:140-142, :338-339
        x = x / torch.sqrt(torch.tensor(float(D)))                             # model.py:144
        mem = self._r(memory.float())
        for i in range(c.num_decoder_layers):
            p = f"decoder.layers.{i}."
            # self-attention block, post-norm
            qkv = self._r(self._r(x) @ w[p + "self_attn.in_proj_weight"].t() + w[p + "self_attn.in_proj_bias"])
            q, k, v = qkv.split(D, dim=-1)
            sa = self._attend(q, k, v, mask) @ w[p + "self_attn.out_proj.weight"].t() + w[p + "self_attn.out_proj.bias"]
            x = self._ln(x + sa, p + "norm1")
            # cross-attention over the frame tokens
            wi, bi = w[p + "multihead_attn.in_proj_weight"], w[p + "multihead_attn.in_proj_bias"]
            q = self._r(self._r(x) @ wi[:D].t() + bi[:D])
            kv = self._r(mem @ wi[D:].t() + bi[D:])
            k, v = kv.split(D, dim=-1)
            ca = self._attend(q, k, v, None) @ w[p + "multihead_attn.out_proj.weight"].t() + w[p + "multihead_attn.out_proj.bias"]
            x = self._ln(x + ca, p + "norm2")
            # feed-forward
            h = self._r(torch.relu(self._r(x) @ w[p + "linear1.weight"].t() + w[p + "linear1.bias"]))
            x = self._ln(x + h @ w[p + "linear2.weight"].t() + w[p + "linear2.bias"], p + "norm3")
        return self._r(x) @ w["linear.weight"].t() + w["linear.bias"]         # model.py:152

    def greedy_decode(self, memory: torch.Tensor, max_len: int = 10, stop: str = "all_sep", return_logits: bool = False):
        """model.py:156-187 given ``memory`` (the image encoder is outside this path)."""
        c = self.cfg
        B = memory.shape[0]
        tgt = torch.full((B, 1), c.cls_token_id, dtype=torch.long)            # model.py:171
        steps = []
        for _ in range(max_len):                                               # model.py:173
            out = self.forward_decoder(tgt, memory)
            steps.append(out[:, -1])
            last = out.argmax(-1)[:, -1:]                                      # model.py:178-180
            tgt = torch.cat([tgt, last], dim=1)                                # model.py:182
            if stop == "all_sep" and bool((last.squeeze(-1) == c.sep_token_id).all()):   # model.py:184
                break
        return (tgt, torch.stack(steps, 1)) if return_logits else tgt


    def beam_search(self, memory: torch.Tensor, max_len: int = 10, k: int = 3) -> torch.Tensor:
        """model.py:189-318 given ``memory``: k beams, no end-of-sequence handling, every beam proposes its
        top-k continuations, the k best of the k*k candidates survive; returns the best beam [B, max_len]."""
        c = self.cfg
        B = memory.shape[0]
        tgt = torch.full((B, 1), c.cls_token_id, dtype=torch.long)
        logp = torch.log_softmax(self.forward_decoder(tgt, memory)[:, -1], dim=-1)        # model.py:221-225
        scores, top = logp.topk(k, dim=-1)
        seqs = torch.cat([tgt.unsqueeze(1).expand(-1, k, -1), top.unsqueeze(-1)], dim=-1)  # model.py:227
        for _ in range(2, max_len):                                                        # model.py:230
            cand, toks = [], []
            for i in range(k):                                                             # model.py:232-249
                lp = torch.log_softmax(self.forward_decoder(seqs[:, i], memory)[:, -1], dim=-1)
                ts, ti = lp.topk(k, dim=-1)
                cand.append(scores[:, i:i + 1] + ts)
                toks.append(ti)
            cand, toks = torch.cat(cand, dim=1), torch.cat(toks, dim=1)                    # [B, k*k], beam-major
            sel = cand.sort(dim=1, descending=True).indices[:, :k]                         # model.py:252-256
            beam = sel // k
            seqs = torch.cat([seqs.gather(1, beam.unsqueeze(-1).expand(-1, -1, seqs.shape[-1])),
                              toks.gather(1, sel).unsqueeze(-1)], dim=-1)                  # model.py:259-275
            scores = cand.gather(1, sel)
        return seqs[torch.arange(B), scores.argmax(dim=-1)]                                # model.py:317


def make_memory(B: int, F: int, D: int, seed: int) -> torch.Tensor:
    """Seeded stand-in for the spatially averaged TinyViT features (model.py:124), O(1) magnitude."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, F, D, generator=g)
