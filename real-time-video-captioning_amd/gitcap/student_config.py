"""Hyper-parameters, weight names and synthetic weights of the STUDENT caption decoder
(SURVEY.md par. 8 row f.2; reference: ``StudentCandidateV1``, /root/reference/src/models/model.py:50-187).

The reference's defaults (config.py:78-83): TinyViT-21M encoder, d_model 576, 8 heads (head_dim 72),
d_ffn 1024, 2 decoder layers, BERT vocabulary (30522, CLS 101, SEP 102, PAD 0).  The decoder is a
``torch.nn.TransformerDecoder`` (post-LN, ReLU, eps 1e-5, batch_first; model.py:82-85) attending
(a) causally over the caption so far, with PAD tokens masked as keys (model.py:134-136,
src/utils/masking.py), and (b) over ``memory`` = one token per frame, the spatial mean of TinyViT's last
feature map (model.py:124).

Weight names are the reference's own ``state_dict()`` keys, so a checkpoint written by the reference's
trainer loads without a key map:

  embed.weight [V, D]                       model.py:87
  pos_enc.pe [1, 500, D]                    model.py:324-335 (registered buffer)
  decoder.layers.{i}.self_attn.in_proj_weight [3D, D] / in_proj_bias [3D] / out_proj.{weight,bias}
  decoder.layers.{i}.multihead_attn.(same four)          cross-attention over memory
  decoder.layers.{i}.linear1.{weight [FF, D], bias}  linear2.{weight [D, FF], bias}
  decoder.layers.{i}.norm1|norm2|norm3.{weight,bias}
  linear.weight [V, D]  linear.bias [V]     model.py:89

The TinyViT image encoder (``timm==0.9.16``, absent from this image) is NOT part of this path: the
decoder takes ``memory [B, F, D]`` (see gitcap/student.py for how a caller plugs an encoder in).
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict

import numpy as np


@dataclass(frozen=True)
class StudentConfig:
    d_model: int = 576
    n_head: int = 8
    d_ffn: int = 1024
    num_decoder_layers: int = 2
    vocab_length: int = 30522
    cls_token_id: int = 101
    sep_token_id: int = 102
    pad_token_id: int = 0          # create_padding_mask default (src/utils/masking.py:4)
    mem_tokens: int = 6            # frames per clip = memory tokens (model.py:124)
    max_pos: int = 500             # PositionalEncoding max_len (model.py:324)
    ln_eps: float = 1e-5           # nn.TransformerDecoderLayer default

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_head

    def validate(self) -> None:
        assert self.d_model % self.n_head == 0
        assert self.d_model % 32 == 0 and self.d_ffn % 32 == 0, "skinny GEMMs step K by 32"
        assert self.d_model % 16 == 0 and self.d_ffn % 16 == 0
        assert self.head_dim % 8 == 0 and self.head_dim <= 128, "attention kernel loads 16-byte head slices"
        assert self.d_model <= 1024


def student_base() -> StudentConfig:
    """config.py:78-83 of the reference."""
    return StudentConfig()


def student_tiny() -> StudentConfig:
    """Small config for parity tests (head_dim 16, odd vocabulary size)."""
    return StudentConfig(d_model=64, n_head=4, d_ffn=96, num_decoder_layers=2, vocab_length=97,
                         cls_token_id=1, sep_token_id=2)


class CStudentConfig(ctypes.Structure):
    """Field order of ``struct gitcap_student_config`` (include/gitcap.h)."""
    _fields_ = [("d_model", ctypes.c_int32), ("n_head", ctypes.c_int32), ("d_ffn", ctypes.c_int32),
                ("num_layers", ctypes.c_int32), ("vocab_size", ctypes.c_int32), ("cls_token_id", ctypes.c_int32),
                ("sep_token_id", ctypes.c_int32), ("pad_token_id", ctypes.c_int32), ("mem_tokens", ctypes.c_int32),
                ("max_pos", ctypes.c_int32), ("max_rows", ctypes.c_int32), ("max_text_len", ctypes.c_int32),
                ("ln_eps", ctypes.c_float)]

    @classmethod
    def from_config(cls, cfg: StudentConfig, max_rows: int, max_text_len: int) -> "CStudentConfig":
        return cls(cfg.d_model, cfg.n_head, cfg.d_ffn, cfg.num_decoder_layers, cfg.vocab_length, cfg.cls_token_id,
                   cfg.sep_token_id, cfg.pad_token_id, cfg.mem_tokens, cfg.max_pos, max_rows, max_text_len, cfg.ln_eps)


def student_shapes(cfg: StudentConfig) -> "OrderedDict[str, tuple]":
    D, FF, V = cfg.d_model, cfg.d_ffn, cfg.vocab_length
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embed.weight"] = (V, D)
    s["pos_enc.pe"] = (1, cfg.max_pos, D)
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.layers.{i}."
        for att in ("self_attn", "multihead_attn"):
            s[p + att + ".in_proj_weight"] = (3 * D, D)
            s[p + att + ".in_proj_bias"] = (3 * D,)
            s[p + att + ".out_proj.weight"] = (D, D)
            s[p + att + ".out_proj.bias"] = (D,)
        s[p + "linear1.weight"] = (FF, D); s[p + "linear1.bias"] = (FF,)
        s[p + "linear2.weight"] = (D, FF); s[p + "linear2.bias"] = (D,)
        for n in ("norm1", "norm2", "norm3"):
            s[p + n + ".weight"] = (D,); s[p + n + ".bias"] = (D,)
    s["linear.weight"] = (V, D)
    s["linear.bias"] = (V,)
    return s


def is_student_gemm_weight(name: str) -> bool:
    """Tensors the device stores as bf16 GEMM operands (everything else stays fp32)."""
    return name.endswith(("in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight")) or name == "linear.weight"


def positional_table(d_model: int, max_len: int = 500) -> np.ndarray:
    """The sin/cos table of model.py:324-335, same fp32 operation order ([1, max_len, d_model])."""
    import torch
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(torch.log(torch.tensor(10000.0)) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy().copy()


def student_synthetic_weights(cfg: StudentConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights (numpy PCG64, independent of torch's RNG): embeddings N(0,1) like
    nn.Embedding, projections N(0, 1/in_features), LayerNorm near (1, 0), small biases."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w: Dict[str, np.ndarray] = {}
    for name, shape in student_shapes(cfg).items():
        if name == "pos_enc.pe":
            w[name] = positional_table(cfg.d_model, cfg.max_pos)
        elif name == "embed.weight":
            w[name] = rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith(("norm1.weight", "norm2.weight", "norm3.weight")):
            w[name] = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("bias"):
            w[name] = (0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:
            w[name] = (rng.standard_normal(shape, dtype=np.float32) / math.sqrt(shape[-1])).astype(np.float32)
    return w


def student_stress_weights(cfg: StudentConfig, seed: int = 0, gamma_gain: float = 8.0, fc1_gain: float = 20.0,
                           qk_gain: float = 3.0) -> Dict[str, np.ndarray]:
    """``student_synthetic_weights`` + the statistics of trained checkpoints (second weight family of the student decoder's parity
    tests, the counterpart of gitcap.weights.stress_weights): LayerNorm gamma x 8 on 4 channels of every ``norm3`` (the layer
    output: the next layer's q|k|v operand, the residual stream, the vocabulary head's operand -- whose columns for the last
    layer's outlier channels are divided by the same factor), two ``linear1`` rows per layer x 20 (ReLU outputs of tens next to
    O(1) ones), q and k of head 0 of both attentions x 3 (scores x 9: peaked softmax rows, also over the 6 memory tokens)."""
    w = {k: v.copy() for k, v in student_synthetic_weights(cfg, seed).items()}
    rng = np.random.Generator(np.random.PCG64([seed, 0x57e55]))
    D, hd = cfg.d_model, cfg.d_model // cfg.n_head
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.layers.{i}."
        oc = rng.choice(D, size=4, replace=False)
        w[p + "norm3.weight"][oc] *= gamma_gain
        if i + 1 == cfg.num_decoder_layers:
            w["linear.weight"][:, oc] /= gamma_gain
        w[p + "linear1.weight"][rng.choice(cfg.d_ffn, size=2, replace=False)] *= fc1_gain
        for att in ("self_attn", "multihead_attn"):
            w[p + att + ".in_proj_weight"][0:hd] *= qk_gain
            w[p + att + ".in_proj_weight"][D:D + hd] *= qk_gain
    return w


def check_student_shapes(cfg: StudentConfig, weights) -> None:
    for name, shape in student_shapes(cfg).items():
        if name not in weights:
            raise KeyError(f"missing student weight {name}")
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(weights[name].shape)} != expected {shape}")
