"""``StudentCaptioner``: the reference's ``StudentCandidateV1`` call surface
(/root/reference/src/models/model.py:50-187) over libgitcap's student-decoder entry points
(include/gitcap.h, csrc/student.hip).  SURVEY.md par. 8 row f.2.

Same constructor keywords (model.py:55-57), ``forward`` / ``forward_image_enc`` /
``forward_decoder`` / ``greedy_decode`` with the reference's argument meaning, ``state_dict`` keys of
the reference.  The decoder (embedding, positional table, 2 x (self-attention, cross-attention over the
frame tokens, FFN), vocabulary head, greedy loop with the all-rows-SEP stop rule) runs in HIP kernels
with an exact KV cache.  The TinyViT image encoder is ``timm`` code that is absent from this image: it
is NOT rebuilt here.  A caller who has it passes it as ``image_encoder`` (any module mapping
``[B*F,3,H,W]`` to the list of feature maps, model.py:117); ``greedy_decode`` also accepts the frame
features ``memory [B, F, d_model]`` directly.  There is no CPU or PyTorch fallback for the decoder.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Mapping, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .student_config import (CStudentConfig, StudentConfig, check_student_shapes, positional_table, student_shapes)

STOP_NEVER, STOP_ALL_SEP = 0, 1


def _rebuild_student(cfg_dict, weights, kwargs):
    return StudentCaptioner(cfg=StudentConfig(**cfg_dict), weights=weights, **kwargs)


class StudentCaptioner(nn.Module):
    def __init__(self, image_enc_name: Optional[str] = None, d_model: int = 576, n_head: int = 8, d_ffn: int = 1024,
                 dropout: float = 0.0, num_decoder_layers: int = 2, vocab_length: int = 30522, cls_token_id: int = 101,
                 sep_token_id: int = 102, *, cfg: Optional[StudentConfig] = None,
                 weights: Optional[Mapping[str, np.ndarray]] = None, image_encoder: Optional[nn.Module] = None,
                 device: str | torch.device = "cuda:0", max_batch: int = 16, max_text_len: int = 32,
                 mem_tokens: int = 6, stop: str = "all_sep"):
        super().__init__()
        if cfg is None:
            cfg = StudentConfig(d_model=d_model, n_head=n_head, d_ffn=d_ffn, num_decoder_layers=num_decoder_layers,
                                vocab_length=vocab_length, cls_token_id=cls_token_id, sep_token_id=sep_token_id,
                                mem_tokens=mem_tokens)
        cfg.validate()
        self.cfg = cfg
        self.image_enc_name = image_enc_name            # kept for callers that log it; no encoder is built from it
        self.image_encoder = image_encoder
        self.cls_token_id, self.sep_token_id = cfg.cls_token_id, cfg.sep_token_id
        self.n_head = cfg.n_head
        self.stop = stop
        self._kw = dict(max_batch=int(max_batch), max_text_len=int(max_text_len), stop=stop)
        self._dev = torch.device(device)
        self._handle = None
        self._weights: Optional[Dict[str, np.ndarray]] = None
        self._lib = _lib.load()                         # raises if libgitcap.so is missing
        self._create()
        if weights is not None:
            self.load_state_dict(weights)

    # ------------------------------------------------------------------ handle management
    def _create(self):
        if self._dev.type != "cuda":
            raise _lib.GitcapError("gitcap runs on an AMD GPU only (no CPU path); got device %s" % self._dev)
        if not torch.cuda.is_available():
            raise _lib.GitcapError("no HIP device visible: gitcap has no CPU fallback")
        self.max_batch, self.max_text_len = self._kw["max_batch"], self._kw["max_text_len"]
        cc = CStudentConfig.from_config(self.cfg, self.max_batch, self.max_text_len)
        h = ctypes.c_void_p()
        idx = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
        self._dev = torch.device("cuda", idx)
        rc = self._lib.gitcap_student_create(ctypes.byref(cc), idx, ctypes.byref(h))
        self._check(None, rc, "gitcap_student_create")
        self._handle = h

    def _check(self, handle, rc, what):
        if rc != 0:
            msg = self._lib.gitcap_student_last_error(handle)
            raise _lib.GitcapError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")

    def _call(self, name, *args):
        self._check(self._handle, getattr(self._lib, name)(self._handle, *args), name)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.gitcap_student_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)

    # ------------------------------------------------------------------ nn.Module surface
    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if isinstance(dev, (str, torch.device)):
            dev = torch.device(dev)
            if dev.type != "cuda":
                raise _lib.GitcapError("gitcap has no CPU path; .to(%s) refused" % dev)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if idx != self._dev.index:
                self._lib.gitcap_student_destroy(self._handle)
                self._dev = torch.device("cuda", idx)
                self._create()
                if self._weights is not None:
                    self._upload(self._weights)
            if self.image_encoder is not None:
                self.image_encoder.to(self._dev)
        return self

    def state_dict(self, *a, **k):
        return {n: torch.from_numpy(v) for n, v in (self._weights or {}).items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        """Takes the reference's checkpoint keys as they are (src/inference.py:38).  Keys outside the decoder
        (image_encoder.*, projectors.*, upsample, project, project_decoder, the unused template
        ``decoder_layer.*``) are ignored; a missing ``pos_enc.pe`` buffer is rebuilt from model.py:324-335."""
        w = {}
        for name, shape in student_shapes(self.cfg).items():
            if name not in state_dict:
                if name == "pos_enc.pe":
                    w[name] = positional_table(self.cfg.d_model, self.cfg.max_pos)
                    continue
                raise KeyError(f"missing student weight {name}")
            v = state_dict[name]
            w[name] = np.ascontiguousarray(v.detach().cpu().float().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        check_student_shapes(self.cfg, w)
        self._upload(w)
        self._weights = w
        return self

    def _upload(self, w):
        with torch.cuda.device(self._dev):
            for name in student_shapes(self.cfg):
                arr = np.ascontiguousarray(w[name], dtype=np.float32)
                shape = (ctypes.c_int64 * arr.ndim)(*arr.shape)
                self._call("gitcap_student_load_tensor", name.encode(), arr.ctypes.data_as(ctypes.c_void_p), shape, arr.ndim)
            self._call("gitcap_student_finalize")

    def __reduce__(self):
        if self.image_encoder is not None:
            raise TypeError("pickle the image encoder separately; StudentCaptioner pickles its decoder only")
        from dataclasses import asdict
        kw = dict(self._kw)
        kw["device"] = str(self._dev)
        return _rebuild_student, (asdict(self.cfg), self._weights, kw)

    # ------------------------------------------------------------------ reference API
    def _memory(self, memory: torch.Tensor) -> torch.Tensor:
        if memory.dim() != 3 or memory.shape[1] != self.cfg.mem_tokens or memory.shape[2] != self.cfg.d_model:
            raise ValueError(f"expected memory [B,{self.cfg.mem_tokens},{self.cfg.d_model}], got {tuple(memory.shape)}")
        if memory.shape[0] == 0 or memory.shape[0] > self.max_batch:
            raise ValueError(f"batch {memory.shape[0]} outside 1..max_batch={self.max_batch}")
        return memory.to(device=self._dev, dtype=torch.float32).contiguous()

    @torch.no_grad()
    def forward_image_enc(self, x: torch.Tensor):
        """model.py:108-126: frames [B,F,C,H,W] -> (feature maps, memory [B,F,De]) through the caller's encoder."""
        if self.image_encoder is None:
            raise _lib.GitcapError("StudentCaptioner was built without an image_encoder: the TinyViT encoder (timm) is not "
                                   "part of libgitcap; pass image_encoder=... or call the decoder with memory [B,F,d_model]")
        s = x.shape
        fmaps = self.image_encoder(x.to(self._dev).view(s[0] * s[1], *s[2:]))
        memory = torch.mean(fmaps[-1], dim=[2, 3]).view(s[0], s[1], -1)
        return fmaps, memory

    @torch.no_grad()
    def forward_decoder(self, y: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
        """model.py:128-154: y [B,T] ids, memory [B,F,D] -> logits [B,T,V] (fp32, on the device)."""
        mem = self._memory(memory)
        ids = y.to(device=self._dev, dtype=torch.int64).contiguous()
        B, T = ids.shape
        if B != mem.shape[0]:
            raise ValueError("y and memory disagree on the batch size")
        if T < 1 or T > self.max_text_len + 1:
            raise ValueError(f"T={T} outside 1..max_text_len+1={self.max_text_len + 1}")
        logits = torch.empty((B, T, self.cfg.vocab_length), dtype=torch.float32, device=self._dev)
        with torch.cuda.device(self._dev):
            self._call("gitcap_student_set_memory", ctypes.c_void_p(mem.data_ptr()), B, self._stream())
            self._call("gitcap_student_forward_decoder", ctypes.c_void_p(ids.data_ptr()), T, B, T,
                       ctypes.c_void_p(logits.data_ptr()), self._stream())
        return logits

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        """model.py:99-106: feature maps + [logits]."""
        fmaps, memory = self.forward_image_enc(x)
        return list(fmaps) + [self.forward_decoder(y, memory)]

    @torch.no_grad()
    def greedy_decode(self, src: torch.Tensor, max_len: int = 10, stop: Optional[str] = None) -> torch.Tensor:
        """model.py:156-187.  ``src``: frames [B,F,C,H,W] (needs ``image_encoder``) or memory [B,F,D].
        Returns int64 [B, 1+steps] starting with CLS, on ``src``'s device."""
        out_dev = src.device
        memory = self.forward_image_enc(src)[1] if src.dim() == 5 else src
        mem = self._memory(memory)
        if max_len < 1 or max_len > self.max_text_len:
            raise ValueError(f"max_len={max_len} outside 1..max_text_len={self.max_text_len}")
        mode = {"all_sep": STOP_ALL_SEP, "never": STOP_NEVER}[stop or self.stop]
        B = mem.shape[0]
        ids = torch.empty((B, max_len + 1), dtype=torch.int64, device=self._dev)
        steps = torch.zeros(1, dtype=torch.int32, device=self._dev)
        with torch.cuda.device(self._dev):
            self._call("gitcap_student_greedy", ctypes.c_void_p(mem.data_ptr()), B, max_len, mode,
                       ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(steps.data_ptr()), self._stream())
        n = int(steps.item()) if mode == STOP_ALL_SEP else max_len
        ids = ids[:, :1 + n]
        return ids.to(out_dev) if out_dev != ids.device else ids

    generate = greedy_decode

    @torch.no_grad()
    def beam_search(self, src: torch.Tensor, max_len: int = 10, k: int = 3) -> torch.Tensor:
        """model.py:189-318: k beams without end-of-sequence handling; returns the best beam [B, max_len].
        Runs on the device with the exact KV cache and no host round trip (``gitcap_student_beam_search``): the k beams
        of a clip are rows b*k+i (B*k <= max_batch), candidates are ranked by the beam top-k kernel, the cached K/V rows
        follow their beams."""
        out_dev = src.device
        memory = self.forward_image_enc(src)[1] if src.dim() == 5 else src
        mem = self._memory(memory)
        B = mem.shape[0]
        if B * k > self.max_batch:
            raise ValueError(f"B*k={B * k} rows > max_batch={self.max_batch}")
        if max_len - 1 > self.max_text_len:
            raise ValueError(f"max_len={max_len} exceeds max_text_len+1={self.max_text_len + 1}")
        if max_len < 2 or k < 1 or k > 16:
            raise ValueError("beam_search needs max_len >= 2 and 1 <= k <= 16")
        best = torch.empty((B, max_len), dtype=torch.int64, device=self._dev)
        with torch.cuda.device(self._dev):
            self._call("gitcap_student_beam_search", ctypes.c_void_p(mem.data_ptr()), B, k, max_len,
                       ctypes.c_void_p(best.data_ptr()), self._stream())
        return best.to(out_dev) if out_dev != best.device else best

    @torch.no_grad()
    def beam_search_host(self, src: torch.Tensor, max_len: int = 10, k: int = 3) -> torch.Tensor:
        """The same search driven from the host the way the reference writes it (every step recomputes the whole prefix
        through ``forward_decoder``, one host sync per step): the cross-check of ``beam_search`` in the tests."""
        out_dev = src.device
        memory = self.forward_image_enc(src)[1] if src.dim() == 5 else src
        mem = self._memory(memory)
        B = mem.shape[0]
        if B * k > self.max_batch:
            raise ValueError(f"B*k={B * k} rows > max_batch={self.max_batch}")
        if max_len - 1 > self.max_text_len:
            raise ValueError(f"max_len={max_len} exceeds max_text_len+1={self.max_text_len + 1}")
        tgt = torch.full((B, 1), self.cls_token_id, dtype=torch.long, device=self._dev)
        logp = torch.log_softmax(self.forward_decoder(tgt, mem)[:, -1], dim=-1)            # model.py:221-225
        scores, top = logp.topk(k, dim=-1)
        seqs = torch.cat([tgt.unsqueeze(1).expand(-1, k, -1), top.unsqueeze(-1)], dim=-1)  # [B, k, 2]
        mem_rep = mem.repeat_interleave(k, dim=0)                                          # row b*k + i = beam i of clip b
        for _ in range(2, max_len):                                                        # model.py:230
            lp = torch.log_softmax(self.forward_decoder(seqs.reshape(B * k, -1), mem_rep)[:, -1], dim=-1)
            ts, ti = lp.view(B, k, -1).topk(k, dim=-1)                                     # each beam's top-k
            cand = (scores.unsqueeze(-1) + ts).view(B, k * k)                              # beam-major, as all_candidates
            sel = cand.sort(dim=1, descending=True).indices[:, :k]                         # model.py:252-256
            beam = sel // k
            seqs = torch.cat([seqs.gather(1, beam.unsqueeze(-1).expand(-1, -1, seqs.shape[-1])),
                              ti.view(B, k * k).gather(1, sel).unsqueeze(-1)], dim=-1)
            scores = cand.gather(1, sel)
        best = seqs[torch.arange(B, device=self._dev), scores.argmax(dim=-1)]              # model.py:317
        return best.to(out_dev) if out_dev != best.device else best
