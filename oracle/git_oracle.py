"""CPU oracle for the GIT captioning hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (real-time-video-captioning_amd/) never does.

PARITY PINNING: the reference tree holds no golden vectors, fixtures or passing tests for this
path and the arithmetic itself lives in the un-vendored, *unpinned*
third-party package ``generativeimage2text`` (microsoft/GenerativeImage2Text, requirements.txt:19
of the reference; call sites /root/reference/src/models/model.py:11-16, :682, :687).  So parity is
UNPINNED BY THE REFERENCE.  The oracle is instead pinned against the only implementation of the
same published model available offline: ``transformers.models.git`` (HF port of MS GIT, v5.15.0),
through the fixtures in tests/golden/ produced by oracle/gen_golden_hf.py.

What this file restates (plain torch fp32, no ``transformers`` import, no reference import):
  * frame batching, temporal-embedding add and token concat   - model.py:372-382
  * ``self.textual(visual_features, caption_tokens)``          - model.py:412-418
  * model hyper-parameters                                     - model.py:681-700
  * greedy stop semantics of the student API                   - model.py:156-187
  * CLIP-ViT tower / linearLn projection / BERT-style decoder with the GIT block mask:
    published GIT algorithm as ported in transformers/models/git/modeling_git.py
    (:73-113 text embeddings, :116-197 attention, :229-290 layer, :343-423 vision embeddings,
     :465-553 vision block, :591-626 tower, :689-699 projection, :796-884 model forward).

``emulate_bf16=True`` reproduces the rounding points of the HIP path (DESIGN.md par. 3):
GEMM operands (weights and the activations fed to a GEMM, incl. q/k/v, softmax probabilities
and GELU outputs) are rounded to bf16, everything else (accumulators, residual stream,
LayerNorm, softmax statistics, embeddings, logits) stays fp32.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as Fn


def _r(x: torch.Tensor, on: bool) -> torch.Tensor:
    """Round-to-nearest-even to bf16 and back to fp32 (identity when emulation is off)."""
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def _q8(x: torch.Tensor) -> torch.Tensor:
    """OCP e4m3 with one power-of-two scale per row (the smallest 2^e with amax / 2^e <= 448; e4m3 x 2^e is a bf16
    value): the activation encoding of the opt-in fp8-MFMA image pass (csrc: act_quant; docs/LAB_NOTEBOOK.md par. 3).  Rounds to
    nearest even, saturates at +-448 x scale (cannot happen: the scale covers the row's amax)."""
    amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))
    s = torch.exp2(e)
    s = torch.where(s * 448.0 < amax, s * 2.0, s)               # log2 rounding at exact powers of two
    return (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s


def _q8_static(x: torch.Tensor, scale: float) -> torch.Tensor:
    """OCP e4m3 with ONE static power-of-two scale, straight from the fp32 value, saturating at +-448 x scale: what the
    device's `compute="fp8_ffn"` epilogues write for the activation operand of the two FFN GEMMs (csrc/common.h:
    pack_fp8x4; csrc/gemm_f8.hip)."""
    return (x / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale


FP8_FFN_SCALE = 1.0 / 16.0      # default static scale of the e4m3 FFN activations (codes cover +-28; csrc/gitcap.hip: f8_scale)


class GitOracle:
    def __init__(self, cfg, weights: Dict[str, np.ndarray], emulate_bf16: bool = False,
                 threads: Optional[int] = None, emulate_fp8_act: bool = False, fp8_scale: float = FP8_FFN_SCALE,
                 emulate_fp8_v: bool = False):
        self.cfg = cfg
        self.bf = bool(emulate_bf16)
        # True (study, oracle/fp8_act_study.py): the activation operand of every image-row GEMM (not the patch embedding, not
        # the text rows) is e4m3 with a per-row power-of-two scale instead of bf16.
        # "ffn" (the device's opt-in compute="fp8_ffn"): only FC1 and FC2 of the image rows, e4m3 with the static scale
        # FP8_FFN_SCALE, rounded straight from fp32.
        self.f8 = emulate_fp8_act if emulate_fp8_act == "ffn" else bool(emulate_fp8_act)
        self.f8_scale = float(fp8_scale)   # "ffn" mode: the device's gitcap_set_fp8_scale
        self.f8_sat = 0                    # "ffn" mode: codes clamped at +-448 so far (the device's gitcap_fp8_saturations)
        # The device's opt-in kv_cache="v_e4m3": the TEXT rows' attention reads the V rows of the IMAGE keys as e4m3 codes with one
        # power-of-two scale per (token, head) (csrc/rowops.hip: kv_quant_v_kernel); K, the text rows' own K/V and the image rows'
        # own attention stay bf16.
        self.f8v = bool(emulate_fp8_v)
        if threads:
            torch.set_num_threads(threads)
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in weights.items():
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).clone()
            # GEMM weights are stored in bf16 on the device; tables/bias/LN stay fp32
            if self.bf and t.ndim == 2 and k not in ("enc.pos", "temporal", "txt.word", "txt.pos"):
                t = _r(t, True)
            self.w[k] = t

    # ------------------------------------------------------------------ primitives
    def _lin(self, x, name, img: bool = False):
        """y = bf16(x) @ bf16(W)^T + b with fp32 accumulation (img: an image-row GEMM, e4m3 activations when emulate_fp8_act)."""
        if img and self.f8 == "ffn":
            if name.endswith("fc1") or name.endswith("fc2"):
                self.f8_sat += int(((x / self.f8_scale).abs() > 448.0).sum())
                xin = _q8_static(x, self.f8_scale)
            else:
                xin = _r(x, self.bf)
        else:
            xin = _q8(_r(x, self.bf)) if (img and self.f8) else _r(x, self.bf)
        return Fn.linear(xin, self.w[name + ".w"], self.w[name + ".b"])

    def _ln(self, x, name, eps):
        return Fn.layer_norm(x, (x.shape[-1],), self.w[name + ".w"], self.w[name + ".b"], eps)

    def _attn(self, q, k, v, klimit: torch.Tensor):
        """q [B,H,Tq,64], k/v [B,H,Tk,64] (already bf16-rounded when emulating);
        klimit [Tq] = number of visible keys per query row."""
        s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
        Tk = k.shape[-2]
        vis = torch.arange(Tk)[None, :] < klimit[:, None]               # [Tq,Tk]
        s = s.masked_fill(~vis, float("-inf"))
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = torch.matmul(_r(p, self.bf), v) / l
        return o

    @staticmethod
    def _heads(x, H):     # [B,T,H*64] -> [B,H,T,64]
        B, T, _ = x.shape
        return x.view(B, T, H, -1).transpose(1, 2)

    @staticmethod
    def _merge(x):        # [B,H,T,64] -> [B,T,H*64]
        B, H, T, d = x.shape
        return x.transpose(1, 2).reshape(B, T, H * d)

    # ------------------------------------------------------------------ encoder (a2,a3)
    def embed_frames(self, frames: torch.Tensor) -> torch.Tensor:
        """frames [B,F,3,H,W] -> the ViT's residual stream after ln_pre, [B*F, N, Dv] (patchify conv k = stride = p without bias,
        CLS, position embedding, ln_pre: modeling_git.py:343-423, :591-610)."""
        cfg = self.cfg
        B, F = frames.shape[:2]
        p, g, Dv = cfg.patch_size, cfg.grid, cfg.enc_width
        x = frames.reshape(B * F, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * F, g * g, 3 * p * p)
        x = torch.matmul(_r(x.float(), self.bf), self.w["enc.patch_w"].t())        # conv k=stride=p, no bias
        cls = self.w["enc.cls"].expand(B * F, 1, Dv)
        x = torch.cat([cls, x], dim=1) + self.w["enc.pos"][None]
        return self._ln(x, "enc.ln_pre", cfg.enc_ln_eps)

    def enc_block(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """One pre-LN CLIP-ViT block (modeling_git.py:465-553) on the residual stream x [nf, N, Dv]."""
        cfg = self.cfg
        Dv, H = cfg.enc_width, cfg.enc_heads
        N = x.shape[1]
        full = torch.full((N,), N)
        pre = f"enc.L{i}."
        h = self._ln(x, pre + "ln1", cfg.enc_ln_eps)
        qkv = _r(self._lin(h, pre + "qkv", True), self.bf)
        q, k, v = (self._heads(t, H) for t in qkv.split(Dv, dim=-1))
        a = self._merge(self._attn(q, k, v, full))
        x = x + self._lin(a, pre + "proj", True)
        h = self._ln(x, pre + "ln2", cfg.enc_ln_eps)
        h = self._lin(h, pre + "fc1", True)
        h = h * torch.sigmoid(1.702 * h)                                       # QuickGELU
        return x + self._lin(h, pre + "fc2", True)

    def enc_post(self, x: torch.Tensor, B: int, F: int) -> torch.Tensor:
        """ln_post + temporal embedding + concat along tokens (model.py:379-382): [B*F, N, Dv] -> [B, F*N, Dv]."""
        cfg = self.cfg
        N, Dv = x.shape[1], cfg.enc_width
        x = self._ln(x, "enc.ln_post", cfg.enc_ln_eps)
        x = x.view(B, F, N, Dv)
        if cfg.num_frames:
            x = x + self.w["temporal"][:F][None, :, None, :]
        return x.reshape(B, F * N, Dv)

    def encode_frames(self, frames: torch.Tensor, taps: Optional[list] = None) -> torch.Tensor:
        """frames [B,F,3,H,W] fp32 NCHW -> visual features [B, F*N, Dv]
        (= ln_post output + temporal embedding, frames concatenated along tokens:
        model.py:378-382).  taps (a list, optional): receives the residual stream entering every block."""
        B, F = frames.shape[:2]
        x = self.embed_frames(frames)
        for i in range(self.cfg.enc_layers):
            if taps is not None:
                taps.append(x)
            x = self.enc_block(i, x)
        return self.enc_post(x, B, F)

    # ------------------------------------------------------------------ projection (a4)
    def project(self, visual: torch.Tensor) -> torch.Tensor:
        return self._ln(self._lin(visual, "vproj", True), "vproj.ln", self.cfg.proj_ln_eps)

    # ------------------------------------------------------------------ text embedding (a5)
    def embed_text(self, ids: torch.Tensor, pos0: int = 0) -> torch.Tensor:
        T = ids.shape[1]
        e = self.w["txt.word"][ids] + self.w["txt.pos"][pos0:pos0 + T][None]
        return self._ln(e, "txt.ln", self.cfg.dec_ln_eps)

    # ------------------------------------------------------------------ decoder (a6,a7)
    def _dec_layer(self, i, x, k_all, v_all, klimit, img: bool = False):
        """x [B,Tq,D] query rows; k_all/v_all [B,H,Tk,64] all visible keys (incl. this block).  img: image rows only."""
        cfg = self.cfg
        pre = f"dec.L{i}."
        D, H = cfg.dec_width, cfg.dec_heads
        qkv = _r(self._lin(x, pre + "qkv", img), self.bf)
        q = self._heads(qkv[..., :D], H)
        a = self._merge(self._attn(q, k_all, v_all, klimit))
        h = self._ln(self._lin(a, pre + "ao", img) + x, pre + "ln1", cfg.dec_ln_eps)
        f = Fn.gelu(self._lin(h, pre + "fc1", img))                                # erf GELU
        return self._ln(self._lin(f, pre + "fc2", img) + h, pre + "ln2", cfg.dec_ln_eps)

    def _kv(self, i, x, img: bool = False):
        cfg = self.cfg
        D, H = cfg.dec_width, cfg.dec_heads
        qkv = _r(self._lin(x, f"dec.L{i}.qkv", img), self.bf)
        return self._heads(qkv[..., D:2 * D], H), self._heads(qkv[..., 2 * D:], H)

    def dec_layer_full(self, i: int, x: torch.Tensor, S_img: int) -> torch.Tensor:
        """One decoder layer over [image ; text] rows x [B, S_img + T, D] with the GIT block mask (the loop body of decoder_full)."""
        rows = torch.arange(x.shape[1])
        klimit = torch.where(rows < S_img, torch.full_like(rows, S_img), rows + 1)
        k, v = self._kv(i, x)
        return self._dec_layer(i, x, k, v, klimit)

    def dec_layer_img(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """One decoder layer over image rows only, x [B, S_img, D] (image rows never see text): the loop body of image_kv."""
        S_img = x.shape[1]
        k, v = self._kv(i, x, True)
        return self._dec_layer(i, x, k, v, torch.full((S_img,), S_img), True)

    def _vq(self, v: torch.Tensor) -> torch.Tensor:
        """Image-prefix V [B,H,S,64] as the text attention sees it (identity unless emulate_fp8_v)."""
        return _q8(v) if self.f8v else v

    def dec_layer_text(self, i: int, x: torch.Tensor, S_img: int) -> torch.Tensor:
        """The TEXT rows of one decoder layer over [image ; text] rows x [B, S_img + T, D] (image keys' V through _vq): [B, T, D]."""
        T = x.shape[1] - S_img
        k, v = self._kv(i, x)
        v = torch.cat([self._vq(v[:, :, :S_img]), v[:, :, S_img:]], dim=2)
        return self._dec_layer(i, x[:, S_img:], k, v, S_img + torch.arange(T) + 1)

    def decoder_full(self, memory: torch.Tensor, ids: torch.Tensor,
                     return_hidden: bool = False):
        """Full (no-cache) pass over [image ; text] with the GIT block mask:
        image->image full, image->text blocked, text->image full, text->text causal
        (model.py:412-418; mask per modeling_git.py:855-869).  memory = projected visual
        features [B,S_img,D]; ids [B,T].  Returns logits [B,T,V]."""
        cfg = self.cfg
        S_img, T = memory.shape[1], ids.shape[1]
        if self.f8v:                  # the text rows see a different V of the image keys than the image rows do: the split form
            assert not return_hidden
            return self.decoder_text(self.image_kv(memory), ids)
        x = torch.cat([memory, self.embed_text(ids)], dim=1)
        rows = torch.arange(S_img + T)
        klimit = torch.where(rows < S_img, torch.full_like(rows, S_img), rows + 1)
        hidden = [x]                  # L+1 entries [B, S_img+T, D]: the stack's input, then each layer's output
        for i in range(cfg.dec_layers):   # (what forward_one_custom stacks as hidden_states, model.py:419-424)
            k, v = self._kv(i, x)
            x = self._dec_layer(i, x, k, v, klimit)
            hidden.append(x)
        logits = self._lin(x[:, S_img:], "head")
        return (logits, hidden) if return_hidden else logits

    # the image half once, any number of text prefixes against it (what a search loop over a no-cache `step` needs:
    # decoder_full recomputes the whole image prefix for every row of every step)
    def image_kv(self, memory: torch.Tensor):
        """Per-layer K/V of the image rows [B,H,S_img,64] (text independent: image rows never see text)."""
        cfg = self.cfg
        S_img = memory.shape[1]
        x, kv = memory, []
        full = torch.full((S_img,), S_img)
        for i in range(cfg.dec_layers):
            k, v = self._kv(i, x, True)
            kv.append((k, self._vq(v)))             # what the TEXT rows read; the image rows' own attention below keeps bf16 V
            if i + 1 < cfg.dec_layers:
                x = self._dec_layer(i, x, k, v, full, True)
        return kv

    def decoder_text(self, image_kv, ids: torch.Tensor, clip_of_row: torch.Tensor | None = None) -> torch.Tensor:
        """Logits [R,T,V] of the text rows for prefixes ids [R,T] given image_kv (from image_kv()); row r uses the image
        K/V of clip clip_of_row[r] (default r).  Same arithmetic as decoder_full restricted to the text rows."""
        cfg = self.cfg
        R, T = ids.shape
        sel = torch.arange(R) if clip_of_row is None else clip_of_row
        S_img = image_kv[0][0].shape[2]
        klimit = S_img + torch.arange(T) + 1
        x = self.embed_text(ids)
        for i in range(cfg.dec_layers):
            kt, vt = self._kv(i, x)
            k = torch.cat([image_kv[i][0][sel], kt], dim=2)
            v = torch.cat([image_kv[i][1][sel], vt], dim=2)
            x = self._dec_layer(i, x, k, v, klimit)
        return self._lin(x, "head")

    # exact KV-cached variant (image K/V are text independent, so the cache is exact)
    def prefill(self, memory: torch.Tensor, ids: torch.Tensor):
        cfg = self.cfg
        S_img, T = memory.shape[1], ids.shape[1]
        if self.f8v:                  # image half once (its V quantised for the text rows), then the text prefix against it
            ikv = self.image_kv(memory)
            x = self.embed_text(ids)
            cache = []
            for i in range(cfg.dec_layers):
                kt, vt = self._kv(i, x)
                k, v = torch.cat([ikv[i][0], kt], dim=2), torch.cat([ikv[i][1], vt], dim=2)
                cache.append([k, v])
                x = self._dec_layer(i, x, k, v, S_img + torch.arange(T) + 1)
            return self._lin(x[:, -1], "head"), cache, T
        x = torch.cat([memory, self.embed_text(ids)], dim=1)
        rows = torch.arange(S_img + T)
        klimit = torch.where(rows < S_img, torch.full_like(rows, S_img), rows + 1)
        cache = []
        for i in range(cfg.dec_layers):
            k, v = self._kv(i, x)
            cache.append([k, v])
            x = self._dec_layer(i, x, k, v, klimit)
        return self._lin(x[:, -1], "head"), cache, T

    def step(self, cache, tok: torch.Tensor, t: int):
        """tok [B] = token at text position t; returns logits [B,V] for position t+1."""
        cfg = self.cfg
        x = self.embed_text(tok[:, None], pos0=t)
        for i in range(cfg.dec_layers):
            k, v = self._kv(i, x)
            cache[i][0] = torch.cat([cache[i][0], k], dim=2)
            cache[i][1] = torch.cat([cache[i][1], v], dim=2)
            Tk = cache[i][0].shape[2]
            x = self._dec_layer(i, x, cache[i][0], cache[i][1], torch.tensor([Tk]))
        return self._lin(x[:, -1], "head")

    # ------------------------------------------------------------------ API restatements
    def forward_image_enc(self, frames):
        visual = self.encode_frames(frames)
        return visual, self.project(visual)

    def forward_output_logits(self, frames, ids):
        """Teacher-forced logits [B,T,V] + visual features (model.py:747-760, batched)."""
        visual, mem = self.forward_image_enc(frames)
        return self.decoder_full(mem, ids), visual

    def greedy_decode(self, frames: torch.Tensor, max_len: int = 20, stop: str = "all_sep",
                      use_cache: bool = True, return_logits: bool = False):
        """Restates StudentCandidateV1.greedy_decode (model.py:156-187) on the GIT arithmetic:
        CLS start [B,1] (:171); <= max_len steps (:173); argmax of the last position (:178-180);
        append (:182); stop iff ALL rows emitted SEP in the same step (:184).  stop='never'
        disables the check (fixed-work benchmarking).  Returns int64 [B, 1+steps]."""
        cfg = self.cfg
        B = frames.shape[0]
        _, mem = self.forward_image_enc(frames)
        ids = torch.full((B, 1), cfg.cls_token_id, dtype=torch.long)
        all_logits: List[torch.Tensor] = []
        cache = None
        for t in range(max_len):
            if use_cache:
                if cache is None:
                    logits, cache, _ = self.prefill(mem, ids)
                else:
                    logits = self.step(cache, ids[:, -1], t)
            else:
                logits = self.decoder_full(mem, ids)[:, -1]
            all_logits.append(logits)
            nxt = logits.argmax(dim=-1, keepdim=True)
            ids = torch.cat([ids, nxt], dim=1)
            if stop == "all_sep" and bool(torch.all(nxt.squeeze(-1) == cfg.sep_token_id)):
                break
        return (ids, torch.stack(all_logits, 1)) if return_logits else ids


def make_frames(B: int, F: int, size: int, seed: int = 1234) -> torch.Tensor:
    """Synthetic CLIP-normalised-looking frames: numpy PCG64 so the same
    tensor is regenerated bit-for-bit on any machine."""
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.standard_normal((B, F, 3, size, size), dtype=np.float32))
