"""Hyper-parameters of the GIT captioning path.

Mirrors the values the reference hard-codes in ``get_git_model``
(/root/reference/src/models/model.py:681-718) and the teacher yaml
(data/teacher_configs/GIT_LARGE_MSRVTT/parameter.yaml:1-3):
vocab 30522, hidden 768, 6 decoder layers, 12 heads, FFN 3072, max caption
length 1024, CLS=101 / SEP=102 / PAD=0 (src/utils/tokenizer.py:5-27).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class GitCapConfig:
    image_size: int = 224
    patch_size: int = 16
    enc_width: int = 768          # Dv
    enc_layers: int = 12
    enc_heads: int = 12
    enc_ffn: int = 3072
    dec_width: int = 768          # D
    dec_layers: int = 6
    dec_heads: int = 12
    dec_ffn: int = 3072
    vocab_size: int = 30522
    max_text_pos: int = 1024
    num_frames: int = 6           # num_image_with_embedding; 0 = single image, no temporal embedding
    enc_ln_eps: float = 1e-5
    dec_ln_eps: float = 1e-12
    proj_ln_eps: float = 1e-5
    cls_token_id: int = 101
    sep_token_id: int = 102
    pad_token_id: int = 0

    # ---- derived -----------------------------------------------------------
    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens_per_frame(self) -> int:      # N (CLS + patches)
        return self.grid * self.grid + 1

    @property
    def patch_dim(self) -> int:             # 3*p*p
        return 3 * self.patch_size * self.patch_size

    def frames(self, F: int | None = None) -> int:
        return max(1, self.num_frames) if F is None else F

    def image_tokens(self, F: int) -> int:  # S_img
        return F * self.tokens_per_frame

    def validate(self) -> None:
        assert self.image_size % self.patch_size == 0
        assert self.enc_width % self.enc_heads == 0 and self.enc_width // self.enc_heads == 64, \
            "attention kernels are specialised for head_dim 64 (CLIP ViT-B/16, ViT-L/14 and the GIT decoder all use 64)"
        assert self.dec_width // self.dec_heads == 64 and self.dec_width % self.dec_heads == 0
        for n in (self.enc_width, self.enc_ffn, self.dec_width, self.dec_ffn):
            assert n % 128 == 0, "GEMM N/K dims must be multiples of 128"

    def to_dict(self) -> dict:
        return asdict(self)


def git_base(num_frames: int = 6) -> GitCapConfig:
    """GIT-base: CLIP ViT-B/16 encoder + 6-layer decoder (model.py:681-700 defaults)."""
    return GitCapConfig(num_frames=num_frames)


def git_large(num_frames: int = 6) -> GitCapConfig:
    """GIT-large teacher: CLIPViT_L_14, visual_feature_size 1024 (parameter.yaml:1-3)."""
    return GitCapConfig(patch_size=14, enc_width=1024, enc_layers=24, enc_heads=16,
                        enc_ffn=4096, num_frames=num_frames)


def git_tiny(num_frames: int = 2) -> GitCapConfig:
    """Small config that satisfies every kernel constraint; used by parity tests."""
    return GitCapConfig(image_size=32, patch_size=8, enc_width=128, enc_layers=2, enc_heads=2,
                        enc_ffn=256, dec_width=128, dec_layers=2, dec_heads=2, dec_ffn=256,
                        vocab_size=197, max_text_pos=64, num_frames=num_frames)


class CGitCapConfig(ctypes.Structure):
    """ctypes twin of ``struct gitcap_config`` in include/gitcap.h (field order must match)."""
    _fields_ = [
        ("image_size", ctypes.c_int32), ("patch_size", ctypes.c_int32),
        ("enc_width", ctypes.c_int32), ("enc_layers", ctypes.c_int32),
        ("enc_heads", ctypes.c_int32), ("enc_ffn", ctypes.c_int32),
        ("dec_width", ctypes.c_int32), ("dec_layers", ctypes.c_int32),
        ("dec_heads", ctypes.c_int32), ("dec_ffn", ctypes.c_int32),
        ("vocab_size", ctypes.c_int32), ("max_text_pos", ctypes.c_int32),
        ("num_frames", ctypes.c_int32),
        ("cls_token_id", ctypes.c_int32), ("sep_token_id", ctypes.c_int32),
        ("pad_token_id", ctypes.c_int32),
        ("enc_ln_eps", ctypes.c_float), ("dec_ln_eps", ctypes.c_float),
        ("proj_ln_eps", ctypes.c_float),
        ("max_batch", ctypes.c_int32),      # clips per call the workspace is sized for
        ("max_frames", ctypes.c_int32),     # frames per clip the workspace is sized for
        ("max_text_len", ctypes.c_int32),   # text positions (CLS + generated) per row
        ("max_beams", ctypes.c_int32),
    ]

    @classmethod
    def from_config(cls, cfg: GitCapConfig, max_batch: int, max_frames: int,
                    max_text_len: int, max_beams: int = 1) -> "CGitCapConfig":
        c = cls()
        for name, _ in cls._fields_:
            if hasattr(cfg, name):
                setattr(c, name, getattr(cfg, name))
        c.max_batch, c.max_frames = max_batch, max_frames
        c.max_text_len, c.max_beams = max_text_len, max_beams
        return c
