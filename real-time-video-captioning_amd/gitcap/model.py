"""Python host side of libgitcap: an ``nn.Module`` with the call surface the reference's
callers use, so it can stand in for the model object of farazali7/real-time-video-captioning.

Mirrored interface (reference file:line):
  * ``greedy_decode(src, max_len)``           src/models/model.py:156-187, called by
                                              src/real_time_inference.py:58, src/inference.py:51
  * ``forward_image_enc(x)``                  src/models/model.py:114-133
  * ``forward_decoder(y, memory)``            src/models/model.py:135-154
  * ``forward(x, y)``                         src/models/model.py:105-112
  * ``forward_output_logits(x, y)``           src/models/model.py:747-760 (teacher)
  * ``beam_search(src, max_len, k)``          src/models/model.py:189-317 (signature)
  * ``eval() / to() / load_state_dict()``     src/inference.py:38-40
  * picklable                                 src/real_time_inference.py:8-9 (torch.load of a whole module)

PyTorch is used for device buffers and the current HIP stream only; every FLOP of the path runs
in the hand-written gfx950 kernels behind the C ABI (include/gitcap.h).
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Dict, Mapping, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .config import CGitCapConfig, GitCapConfig, git_base
from . import weights as W

STOP_NEVER, STOP_ALL_SEP = 0, 1


class _Submission:
    """One gitcap_greedy_submit / gitcap_beam_search_submit (possibly several coalesced caller batches); keeps its buffers
    alive until every future attached to it has delivered its result (the frames are what a poisoned submission is re-run from)."""

    def __init__(self, ticket, frames, outs, coalesced, users=1, parts=None):
        self.ticket, self.frames, self.outs, self.coalesced, self.users = ticket, frames, outs, coalesced, users
        self.parts = parts           # host-fed submissions: the callers' CPU tensors (frames is then their list; a re-run reads them)
        self.waited = False
        self.done_ev = None          # recorded on the stream that waited: fires once the submission's last launch has finished
        self.poisoned = False        # in flight when the LayerNorm statistics exchange failed: results undefined, re-run
        self.owner = None            # the model: forgets the submission once every future has delivered

    def rows(self, r0, r1):
        """Frames of caller rows [r0, r1) (what a re-run of a poisoned submission decodes again)."""
        if self.parts is None:
            return self.frames[r0:r1]
        at = 0
        for p in self.parts:
            if at == r0 and at + p.shape[0] == r1:
                return p
            at += p.shape[0]
        return torch.cat(list(self.parts), 0)[r0:r1]

    def release(self):
        self.users -= 1
        if self.users <= 0:
            self.frames = self.parts = None
            if self.owner is not None:
                self.owner._undelivered.discard(self)


class _StagingRing:
    """Host-fed submissions (the reference's callers hold CPU tensors: src/real_time_inference.py:39-58 OpenCV frames,
    src/inference.py:45-51 a DataLoader batch): a ring of pinned host buffers + device buffers, so that the host -> device copy
    of batch i + 1 runs under the compute of batch i.  The copy is enqueued on the CALLER's current stream and the submission is
    made behind it (include/gitcap.h: gitcap_greedy_submit orders the image pass behind the work already on `stream`): the
    caller's stream carries nothing else but the waits for earlier results, so the copy starts at once, and only the image pass
    waits for it.  (A stream of the ring's own -- own_stream=True, GITCAP_COPY_STREAM=own -- is a FIFTH stream beside the
    caller's, the encoder's and the two decode streams; the runtime multiplexes streams onto four hardware queues, the copy then
    queues behind a token loop and every image pass starts late: 1 618 instead of 2 004 captions/s host-fed,
    profiles/r06_host_fed_copy_stream.txt -- the same effect as the fifth stream of round 3's two-encoder experiment.)

    Pageable sources are copied into the pinned buffer by gitcap_host_copy (up to 8 threads, sized to the CPU quota: ATen's own
    parallel copy sizes its pool to the whole machine and takes 20 ms per batch under a 16-core quota); page-locked sources
    (DataLoader(pin_memory=True), a capture ring) are copied from where they are -- asynchronously: the caller must leave such a
    buffer alone until the submission's result() (the usual contract of a non_blocking copy; the future keeps a reference).

    Depth = the library's four slots.  An entry is reused four submissions later; before its device buffer is overwritten the
    copy's stream waits for the submission that read it (its `done_ev`, or the library's own wait while it is still in flight),
    and the host waits for the entry's previous host -> device copy before touching the pinned buffer."""
    DEPTH = 4

    def __init__(self, dev, lib, own_stream=False):
        self.dev, self.lib = dev, lib
        self.stream = torch.cuda.Stream(device=dev) if own_stream else None      # None: the caller's current stream (see above)
        self.entries = [dict(pinned=None, device=None, ev=None, sub=None) for _ in range(self.DEPTH + 1)]   # [-1]: synchronous calls
        self.n = 0

    def copy_stream(self):
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.dev)

    def _host_copy(self, dst, src):
        """src (CPU tensor, any memory) -> dst (a view of the pinned buffer, same shape)."""
        if src.dtype == dst.dtype and src.is_contiguous():
            nbytes = src.numel() * src.element_size()
            rc = self.lib.gitcap_host_copy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), nbytes)
            if rc != 0:
                raise _lib.GitcapError(f"gitcap_host_copy failed (status {rc})")
        else:
            dst.copy_(src)                            # strided or another dtype: ATen's copy converts

    @staticmethod
    def _grow(e, nbytes, dev):
        if e["pinned"] is None or e["pinned"].numel() < nbytes:
            cap = (nbytes + (1 << 20) - 1) >> 20 << 20
            e["pinned"] = torch.empty((cap,), dtype=torch.uint8, pin_memory=True)
            e["device"] = torch.empty((cap,), dtype=torch.uint8, device=dev)

    def stage(self, model, parts, dtype, stream=None):
        """Copy the CPU tensors `parts` (same trailing shape; concatenated along dim 0) to the device.  stream=None: the
        pipelined form (next ring entry, the ring's copy stream = the caller's current stream by default); else the synchronous
        form on that stream, in the entry of its own.
        Returns (device tensor, entry, the stream the copy was enqueued on)."""
        sync = stream is not None
        e = self.entries[-1] if sync else self.entries[self.n % self.DEPTH]
        if not sync:
            self.n += 1
        st = stream if sync else self.copy_stream()
        rows = sum(p.shape[0] for p in parts)
        shape = (rows,) + tuple(parts[0].shape[1:])
        nbytes = rows * int(np.prod(parts[0].shape[1:])) * torch.empty((), dtype=dtype).element_size()
        if e["ev"] is not None:
            e["ev"].synchronize()                     # the pinned buffer's previous host -> device copy
        prev = e["sub"]
        if prev is not None:                          # the device buffer's previous reader (an image pass)
            if prev.done_ev is not None:
                st.wait_event(prev.done_ev)
            elif not prev.poisoned and prev in model._inflight:
                try:
                    model._call("gitcap_greedy_wait", prev.ticket, ctypes.c_void_p(st.cuda_stream))
                except _lib.GitcapExchangeTimeout:    # (the device was drained when the failure was taken)
                    prev.poisoned = True
            e["sub"] = None
        self._grow(e, nbytes, self.dev)
        dview = e["device"][:nbytes].view(dtype).view(shape)
        direct = len(parts) == 1 and parts[0].is_pinned() and parts[0].dtype == dtype and parts[0].is_contiguous()
        if direct:
            hsrc = parts[0]                           # already page-locked (DataLoader(pin_memory=True), a capture ring): no staging copy
        else:
            hsrc = e["pinned"][:nbytes].view(dtype).view(shape)
            r0 = 0
            for p in parts:
                self._host_copy(hsrc[r0:r0 + p.shape[0]], p)
                r0 += p.shape[0]
        with torch.cuda.stream(st):
            dview.copy_(hsrc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        e["ev"] = ev
        return dview, e, st


class _Future:
    """Common part of the result handles: wait, vouch for the result where a host synchronisation makes that possible, and
    re-run the batch on the (by then degraded) handle if the submission was poisoned by a failed statistics exchange."""

    def _deliver(self, extract, synced, rerun):
        m, sub = self._m, self._sub
        m._wait_submission(sub)
        out = None
        if not sub.poisoned:
            out = extract()
            if synced:                         # the data is on the host (or a .item() went through): the flag is meaningful now
                try:
                    m.poll_errors()
                except _lib.GitcapExchangeTimeout:
                    m._poison_inflight()
                    sub.poisoned = True
        if sub.poisoned:
            out = rerun()
        sub.release()
        return out


class CaptionFuture(_Future):
    """Result handle of greedy_decode_async."""

    def __init__(self, model, frames, mode, max_len, out_device):
        self._m, self._frames, self._mode, self._max_len, self._out_device = model, frames, mode, max_len, out_device
        self._sub, self._r0, self._r1 = None, 0, 0
        self._done = None

    def _attach(self, sub, r0, r1):
        self._sub, self._r0, self._r1 = sub, r0, r1
        self._frames = None                      # the submission holds the (concatenated) frames now

    def result(self) -> torch.Tensor:
        if self._done is not None:
            return self._done
        m = self._m
        if self._sub is None:                    # still waiting for partners to coalesce with: run now
            m._flush_pending()
        sub = self._sub

        def extract():
            ids = sub.outs[0][self._r0:self._r1]
            if self._mode == STOP_ALL_SEP:
                if sub.coalesced:                    # reference rule over THIS caller batch (model.py:184)
                    all_sep = (ids[:, 1:] == m.sep_token_id).all(dim=0)
                    nz = torch.nonzero(all_sep)
                    n = int(nz[0].item()) + 1 if nz.numel() else self._max_len
                else:
                    n = int(sub.outs[1].item())
                ids = ids[:, :1 + n]
            return ids.to(self._out_device) if self._out_device != ids.device else ids

        def rerun():
            ids = m._greedy_decode(sub.rows(self._r0, self._r1), self._max_len, self._mode)
            return ids.to(self._out_device) if self._out_device != ids.device else ids

        synced = self._mode == STOP_ALL_SEP or self._out_device.type == "cpu"
        self._done = self._deliver(extract, synced, rerun)
        return self._done


class InferFuture(_Future):
    """Result handle of infer_async: ``result()`` returns the dict of ``GitCaptioner.infer``."""

    def __init__(self, model, sub, kw, out_device, save_logits):
        self._m, self._sub, self._kw, self._out_device, self._save = model, sub, kw, out_device, save_logits
        self._done = None

    def result(self) -> dict:
        if self._done is not None:
            return self._done
        m, sub = self._m, self._sub

        def extract():
            decoded, logprobs, steps, vis = sub.outs
            dev = self._out_device
            mv = lambda t: t if t is None or t.device == dev else t.to(dev)
            return {"predictions": mv(decoded), "logprobs": mv(logprobs[:, None]),
                    "logits_dict": [] if steps is None else steps, "visual_features": mv(vis)}

        def rerun():
            r = m._infer_device(m._to_device(sub.rows(0, sub.outs[0].shape[0])), sync=True, save_logits=self._save,
                                want_visual=sub.outs[3] is not None, **self._kw)
            sub.outs = r
            return extract()

        self._done = self._deliver(extract, self._out_device.type == "cpu", rerun)
        return self._done


def _rebuild(cfg_dict, weights, kwargs):
    return GitCaptioner(GitCapConfig(**cfg_dict), weights, **kwargs)


class GitCaptioner(nn.Module):
    def __init__(self, cfg: Optional[GitCapConfig] = None, weights: Optional[Mapping[str, np.ndarray]] = None, *,
                 device: str | torch.device = "cuda:0", max_batch: int = 16, max_frames: Optional[int] = None,
                 max_text_len: int = 32, max_beams: int = 1, tokenizer=None, stop: str = "all_sep",
                 weight_dtype: str = "bf16", compute: str = "bf16", fp8_scale: Optional[float] = None, kv_cache: str = "bf16",
                 # constructor kwargs of the reference student (model.py:55-57); only the ids/vocab matter here
                 vocab_length: Optional[int] = None, cls_token_id: Optional[int] = None,
                 sep_token_id: Optional[int] = None, **_ignored_student_kwargs):
        super().__init__()
        cfg = cfg or git_base()
        over = {}
        if vocab_length is not None:
            over["vocab_size"] = int(vocab_length)
        if cls_token_id is not None:
            over["cls_token_id"] = int(cls_token_id)
        if sep_token_id is not None:
            over["sep_token_id"] = int(sep_token_id)
        if over:
            cfg = GitCapConfig(**{**cfg.to_dict(), **over})
        cfg.validate()
        self.cfg = cfg
        self.tokenizer = tokenizer                      # attribute the reference's callers read (inference.py:43)
        self.cls_token_id, self.sep_token_id = cfg.cls_token_id, cfg.sep_token_id
        self.stop = stop
        if weight_dtype not in ("bf16", "fp8_e4m3"):
            raise ValueError("weight_dtype must be 'bf16' or 'fp8_e4m3'")
        self.weight_dtype = weight_dtype
        if compute not in ("bf16", "fp8_ffn"):
            raise ValueError("compute must be 'bf16' or 'fp8_ffn'")
        if compute == "fp8_ffn" and weight_dtype != "fp8_e4m3":
            raise ValueError("compute='fp8_ffn' needs weight_dtype='fp8_e4m3' (the fp8 GEMMs read the e4m3 codes as stored)")
        self.compute = compute
        if fp8_scale is not None and compute != "fp8_ffn":
            raise ValueError("fp8_scale is the activation scale of compute='fp8_ffn'")
        self.fp8_scale = None if fp8_scale is None else float(fp8_scale)
        if kv_cache not in ("bf16", "v_e4m3"):
            raise ValueError("kv_cache must be 'bf16' or 'v_e4m3' (the token loop reads the image prefix's V rows as e4m3 codes)")
        self.kv_cache = kv_cache
        self._kw = dict(max_batch=max_batch, max_frames=max_frames, max_text_len=max_text_len, max_beams=max_beams, stop=stop,
                        weight_dtype=weight_dtype, compute=compute, fp8_scale=fp8_scale, kv_cache=kv_cache)
        self._dev = torch.device(device)
        self._handle = None
        self._weights: Optional[Dict[str, np.ndarray]] = None
        self._last_memory = None
        self._pending, self._pending_key = [], None
        self._inflight = []                             # submissions whose wait has not been enqueued yet (<= 4)
        self._undelivered = set()                       # waited for, but a future has still to hand out (or re-run) its rows
        self._ring = None                               # _StagingRing, made when the first CPU tensor arrives
        self._copy_stream = os.environ.get("GITCAP_COPY_STREAM", "caller")  # "caller" | "own" (A/B switch; see _StagingRing)
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
        kw = self._kw
        self.max_batch = int(kw["max_batch"])
        self.max_frames = int(kw["max_frames"] or max(1, self.cfg.num_frames))
        self.max_text_len = int(kw["max_text_len"])
        self.max_beams = int(kw["max_beams"])
        cc = CGitCapConfig.from_config(self.cfg, self.max_batch, self.max_frames, self.max_text_len, self.max_beams)
        h = ctypes.c_void_p()
        idx = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
        self._dev = torch.device("cuda", idx)
        rc = self._lib.gitcap_create(ctypes.byref(cc), idx, ctypes.byref(h))
        _lib.check(self._lib, None, rc, "gitcap_create")
        self._handle = h
        if self.weight_dtype == "fp8_e4m3":     # GEMM weights live in HBM as e4m3 + per-row 2^k scale (half the bytes)
            self._call("gitcap_set_weight_storage", 1)
        if self.compute == "fp8_ffn":           # FC1 / FC2 of the image rows on fp8 MFMA (include/gitcap.h: gitcap_set_compute)
            self._call("gitcap_set_compute", 1)
            if self.fp8_scale is not None:
                self._call("gitcap_set_fp8_scale", ctypes.c_float(self.fp8_scale))
        if self.kv_cache == "v_e4m3":           # image-prefix V as e4m3 codes for the token loop (include/gitcap.h: gitcap_set_kv_cache)
            self._call("gitcap_set_kv_cache", 1)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.gitcap_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._handle, *args)
        if rc == _lib.ERR_EXCHANGE:
            # whichever call learns of a failed statistics exchange: every submission whose rows have not been handed out yet
            # (in flight, or already stream-waited by a synchronous call but not delivered) holds undefined ids
            self._poison_inflight()
        _lib.check(self._lib, self._handle, rc, name)

    def _staging(self):
        if self._ring is None:
            self._ring = _StagingRing(self._dev, self._lib, own_stream=self._copy_stream == "own")
        return self._ring

    def _submit(self, name, *args):
        """A gitcap_*_submit call.  If the library reports a failed statistics exchange at this entry (GITCAP_ERR_EXCHANGE, once)
        nothing was submitted: everything already in flight is undefined (marked; each future re-runs its batch when asked) and the
        call is repeated on the handle, which has switched to the unfused launches."""
        try:
            self._call(name, *args)
        except _lib.GitcapExchangeTimeout:
            self._poison_inflight()
            self._call(name, *args)

    def poll_errors(self):
        """Raises GitcapExchangeTimeout if a fused GEMM + LayerNorm launch gave up waiting since the last check
        (gitcap_poll_errors, include/gitcap.h).  Meaningful after the stream that produced a result was synchronised."""
        self._call("gitcap_poll_errors")

    def _vouched(self, produce):
        """Run `produce()` (which must end in a host synchronisation, e.g. a copy to the CPU) and vouch for its result:
        if the exchange flag was raised meanwhile -- at entry, for earlier work, or by this very call -- the library has
        switched to the unfused launches and the call is repeated once on those."""
        try:
            out = produce()
            self.poll_errors()
            return out
        except _lib.GitcapExchangeTimeout:
            self._poison_inflight()          # their ids are undefined: each future re-runs its batch when asked for its result
            out = produce()
            self.poll_errors()
            return out

    # ------------------------------------------------------------------ submissions in flight
    def _wait_submission(self, sub):
        """Make the current stream wait for a submission (once).  Its frames/ids/steps buffers are owned by the
        model-side table until then: dropping a future early cannot hand them back to the allocator while the
        library's streams still read frames / write ids."""
        if not sub.waited and not sub.poisoned:
            try:
                with torch.cuda.device(self._dev):
                    self._call("gitcap_greedy_wait", sub.ticket, self._stream())
                    sub.done_ev = torch.cuda.Event()
                    sub.done_ev.record(torch.cuda.current_stream(self._dev))
                sub.waited = True
            except _lib.GitcapExchangeTimeout:   # raised now, or this ticket was in flight when it was raised: everything
                self._poison_inflight()          # submitted so far is undefined (the C ABI marks the same tickets)
                sub.poisoned = True
        if sub in self._inflight:
            self._inflight.remove(sub)
            if sub.users > 0:                    # until every future has delivered, a failure reported later still marks it
                sub.owner = self
                self._undelivered.add(sub)

    def _poison_inflight(self):
        for sub in self._inflight:
            sub.poisoned = True
        for sub in self._undelivered:            # stream-waited (e.g. by a synchronous call's _drain) but not handed out yet
            sub.poisoned = True

    def _drain(self):
        """Before a synchronous call: submit what is still waiting to coalesce and order the current stream behind
        every submission in flight (the C ABI orders the device work as well; this releases the buffers)."""
        if self._pending:
            self._flush_pending()
        for sub in list(self._inflight):
            self._wait_submission(sub)

    # ------------------------------------------------------------------ nn.Module surface
    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if isinstance(dev, (str, torch.device)):
            dev = torch.device(dev)
            if dev.type != "cuda":
                raise _lib.GitcapError("gitcap has no CPU path; .to(%s) refused" % dev)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if idx != self._dev.index:
                self._lib.gitcap_destroy(self._handle)
                self._dev = torch.device("cuda", idx)
                self._create()
                if self._weights is not None:
                    self._upload(self._weights)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def state_dict(self, *a, **k):
        return {n: torch.from_numpy(v) for n, v in (self._weights or {}).items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts canonical names (gitcap/weights.py), a transformers GitForCausalLM state dict or the
        MS GenerativeImage2Text checkpoint layout the reference loads (model.py:736-738)."""
        keys = state_dict.keys()
        if "enc.patch_w" in keys:
            w = {k: np.ascontiguousarray(v.detach().cpu().float().numpy() if hasattr(v, "detach") else v,
                                         dtype=np.float32) for k, v in state_dict.items()}
            W.check_shapes(self.cfg, w)
        elif any(k.startswith("git.") for k in keys):
            w = W.from_hf_state_dict(self.cfg, state_dict)
        elif any(k.startswith("image_encoder.") for k in keys):
            w = W.from_ms_state_dict(self.cfg, state_dict)
        else:
            raise KeyError("unrecognised checkpoint layout (expected canonical, HF-GIT or MS-GIT keys)")
        if self.weight_dtype == "fp8_e4m3":      # BASELINE configs[4]: fp8 (e4m3) weight values, per-row 2^k scale
            w = W.quantize_weights_fp8(w)
        self._upload(w)
        self._weights = w
        return self

    def _upload(self, w: Mapping[str, np.ndarray]):
        with torch.cuda.device(self._dev):
            for name in W.canonical_shapes(self.cfg):
                arr = np.ascontiguousarray(w[name], dtype=np.float32)
                shape = (ctypes.c_int64 * arr.ndim)(*arr.shape)
                self._call("gitcap_load_tensor", name.encode(), arr.ctypes.data_as(ctypes.c_void_p), shape, arr.ndim)
            self._call("gitcap_finalize_weights")

    def __reduce__(self):
        kw = dict(self._kw)
        kw["device"] = str(self._dev)
        return _rebuild, (self.cfg.to_dict(), self._weights, kw)

    # ------------------------------------------------------------------ helpers
    def _check_frames(self, x: torch.Tensor):
        """Shape checks only (nothing is moved): -> (5-D view, raw) with raw = uint8 BGR camera frames [B,F,H,W,3] (OpenCV layout,
        real_time_inference.py:39; [F,H,W,3] = one clip) as opposed to transformed frames [B,F,3,S,S] ([B,3,S,S] = one frame each)."""
        raw = x.dtype == torch.uint8
        if raw:
            if x.dim() == 4:
                x = x.unsqueeze(0)
            if x.dim() != 5 or x.shape[-1] != 3:
                raise ValueError(f"expected uint8 frames [B,F,H,W,3], got {tuple(x.shape)}")
            if min(x.shape[2], x.shape[3]) < 1:
                raise ValueError("empty frames")
        else:
            if x.dim() == 4:                                  # [B,3,H,W] single image -> one frame
                x = x.unsqueeze(1)
            if x.dim() != 5 or x.shape[2] != 3 or x.shape[3] != self.cfg.image_size or x.shape[4] != self.cfg.image_size:
                raise ValueError(f"expected frames [B,F,3,{self.cfg.image_size},{self.cfg.image_size}], got {tuple(x.shape)}")
        if x.shape[0] == 0:
            raise ValueError("empty batch")
        if x.shape[1] > self.max_frames:
            raise ValueError(f"{x.shape[1]} frames per clip > max_frames={self.max_frames}")
        return x, raw

    def _to_device(self, x: torch.Tensor) -> torch.Tensor:
        """Checked frames -> contiguous, 16-byte aligned device tensor (fp32 NCHW or uint8 HWC) for a SYNCHRONOUS call.  A CPU
        tensor (real_time_inference.py:57) goes through the pinned staging entry of the synchronous calls on the current stream
        (a pageable x.to(device) is a blocking, runtime-staged copy)."""
        dtype = torch.uint8 if x.dtype == torch.uint8 else torch.float32
        if x.device.type == "cpu":
            with torch.cuda.device(self._dev):
                dv, _, _ = self._staging().stage(self, [x], dtype, stream=torch.cuda.current_stream(self._dev))
            return dv
        x = x.to(device=self._dev, dtype=dtype).contiguous()
        if x.data_ptr() % 16:
            x = x.clone()
        return x

    def _frames(self, x: torch.Tensor) -> torch.Tensor:
        x, raw = self._check_frames(x)
        if raw:
            raise ValueError("this call takes transformed fp32 frames [B,F,3,S,S], not uint8 camera frames")
        return self._to_device(x)

    def _raw_frames(self, x: torch.Tensor) -> torch.Tensor:
        """uint8 BGR camera frames [B,F,H,W,3] (OpenCV layout, real_time_inference.py:39) or [F,H,W,3] for one clip."""
        x, raw = self._check_frames(x)
        if not raw:
            raise ValueError(f"expected uint8 frames [B,F,H,W,3], got {x.dtype}")
        return self._to_device(x)

    def _ids(self, y: torch.Tensor) -> torch.Tensor:
        return y.to(device=self._dev, dtype=torch.int64).contiguous()

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward_image_enc(self, x: torch.Tensor):
        """-> ([], memory) with memory = visual features [B, F*N, Dv] (ln_post + temporal embedding,
        frames concatenated along tokens, model.py:378-382).  Also leaves the decoder's image K/V
        in the handle."""
        self._drain()
        raw = x.dtype == torch.uint8
        fr = self._raw_frames(x) if raw else self._frames(x)
        B, F = fr.shape[:2]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch={self.max_batch} (use greedy_decode for automatic chunking)")
        vis = torch.empty((B, F * self.cfg.tokens_per_frame, self.cfg.enc_width), dtype=torch.float32, device=self._dev)
        with torch.cuda.device(self._dev):
            if raw:
                self._call("gitcap_encode_raw", ctypes.c_void_p(fr.data_ptr()), B, F, fr.shape[2], fr.shape[3],
                           ctypes.c_void_p(vis.data_ptr()), self._stream())
            else:
                self._call("gitcap_encode", ctypes.c_void_p(fr.data_ptr()), B, F, ctypes.c_void_p(vis.data_ptr()), self._stream())
        self._remember_memory(vis)
        return [], vis

    def _remember_memory(self, t: torch.Tensor):
        # identity (a weak reference that dies with the tensor) + version: a recycled address cannot alias it
        self._last_memory = (weakref.ref(t), t._version)

    def _is_last_memory(self, t: torch.Tensor) -> bool:
        lm = self._last_memory
        return lm is not None and lm[0]() is t and lm[1] == t._version

    @torch.no_grad()
    def forward_decoder(self, y: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
        """Teacher-forced logits [B,T,V] for token prefixes y [B,T] given `memory` from
        forward_image_enc (block mask: text->image full, text->text causal)."""
        self._drain()
        ids = self._ids(y)
        B, T = ids.shape
        with torch.cuda.device(self._dev):
            if not self._is_last_memory(memory):
                mem = memory.to(device=self._dev, dtype=torch.float32).contiguous()
                self._call("gitcap_set_visual", ctypes.c_void_p(mem.data_ptr()), mem.shape[0], mem.shape[1], self._stream())
                self._remember_memory(memory)
            logits = torch.empty((B, T, self.cfg.vocab_size), dtype=torch.float32, device=self._dev)
            self._call("gitcap_text_forward", ctypes.c_void_p(ids.data_ptr()), T, B, 1, 0, T,
                       ctypes.c_void_p(logits.data_ptr()), 1, None, 0, self._stream())
        return logits

    def forward(self, x: torch.Tensor, y: Optional[torch.Tensor] = None):
        """``forward(x, y)``: teacher-forced logits [B,T,V] (student signature, model.py:99-106).
        ``forward(x)``: the teacher's form (``GenerativeImageTextTeacher.forward``, model.py:762-793)."""
        if y is None:
            return self.teacher_forward(x)
        _, memory = self.forward_image_enc(x)
        return self.forward_decoder(y, memory)

    @torch.no_grad()
    def teacher_forward(self, x: torch.Tensor, beam_size: int = 4, max_steps: int = 15, on_device: Optional[bool] = None) -> list:
        """``GenerativeImageTextTeacher.forward`` (model.py:762-793): one dict per clip with the keys of
        ``infer`` (model.py:456-461) plus ``cap`` (decoded caption, :770) and ``output`` [1, n, V] = for each of
        the first n predicted words the logits of the beam that scores that word highest (:771-788).
        The reference runs one clip at a time (:765); here the clips go through batched searches (chunks of max_batch, pipelined:
        the image pass of one chunk overlaps the search of the one before) and only the per-clip bookkeeping is a loop.  Default:
        the device-resident search with its per-step logits kept on the device (``infer_async``); on_device=False: the host-side
        operator (one host sync per step).  ``logits_dict`` holds every step's [beams, V] logits; the device search runs all
        max_steps - 1 steps where the reference stops once the clip is done (:640) -- ``output`` only reads the first n.
        Without a tokenizer ``cap`` is None and n counts the tokens before SEP."""
        if on_device is None:
            on_device = beam_size * 2 <= 16
        fr, _ = self._check_frames(x)             # a CPU tensor stays where it is: every chunk is staged as it is submitted
        out = []

        def finish(pred, pred_h, logprobs, steps, vis):       # per-clip bookkeeping of one chunk; steps: [S, Bc * beams, V] on the device
            for b in range(pred.shape[0]):
                dist_all = steps[:, b * beam_size:(b + 1) * beam_size]              # [S, beams, V]
                ids = pred_h[b].tolist()
                if self.tokenizer is not None:
                    cap = self.tokenizer.decode(ids, skip_special_tokens=True)
                    n = min(len(cap.split(" ")), dist_all.shape[0])                 # model.py:771
                else:
                    cap = None
                    body = ids[1:]
                    n = min(body.index(self.sep_token_id) if self.sep_token_id in body else len(body), dist_all.shape[0])
                n = max(n, 1)
                dist = dist_all[:n]                                                 # [n, beams, V]
                words = pred[b, 1:n + 1].to(self._dev)
                at_word = torch.gather(dist, 2, words[:, None, None].expand(-1, beam_size, -1)).squeeze(-1)   # [n, beams]
                best = at_word.argmax(dim=1)                                        # model.py:785
                output = torch.gather(dist, 1, best[:, None, None].expand(-1, -1, dist.shape[-1])).squeeze(1)[None]
                out.append({"predictions": pred[b:b + 1], "logprobs": logprobs[b:b + 1],
                            "logits_dict": [a for a in dist_all.cpu().numpy()],
                            "visual_features": vis[b:b + 1], "output": output, "cap": cap})

        if on_device:
            # a sliding window of submissions, each consumed (and its [max_steps - 1, B * beams, V] logits tensor dropped) before
            # the next is made: device memory is bounded by the window, not by the number of chunks
            kw = dict(beam_size=beam_size, max_steps=max_steps, save_logits=True, visual_features=True)
            futs = []

            def take(f, b0):
                r = f.result()
                pred_h = r["predictions"].cpu()           # a host synchronisation: the result can be vouched for now
                try:
                    self.poll_errors()
                except _lib.GitcapExchangeTimeout:        # (the failure poisoned what is in flight; this chunk is re-run here)
                    r = self.infer_async(fr[b0:b0 + self.max_batch], **kw).result()
                    pred_h = r["predictions"].cpu()
                    self.poll_errors()
                finish(r["predictions"], pred_h, r["logprobs"], r["logits_dict"], r["visual_features"])

            for b0 in range(0, fr.shape[0], self.max_batch):
                if len(futs) >= 3:
                    take(*futs.pop(0))
                futs.append((self.infer_async(fr[b0:b0 + self.max_batch], **kw), b0))
            while futs:
                take(*futs.pop(0))
        else:
            for b0 in range(0, fr.shape[0], self.max_batch):
                r = self.infer(fr[b0:b0 + self.max_batch], beam_size=beam_size, max_steps=max_steps, save_logits=True, on_device=False)
                steps = torch.from_numpy(np.stack([np.asarray(st) for st in r["logits_dict"]])).to(self._dev)
                finish(r["predictions"], r["predictions"].cpu(), r["logprobs"], steps, r["visual_features"])
        return out

    @torch.no_grad()
    def forward_output_logits(self, x: torch.Tensor, y: torch.Tensor, output_hidden_states: bool = False):
        """Teacher API (model.py:747-760): per clip lists of logits [1,T,V], visual features [1,F*N,Dv] and hidden
        states; computed as ONE batch instead of the reference's clip-by-clip loop (:752-759).
        The third list (model.py:419-424: the decoder stack's per-layer hidden states over [image ; text], stacked
        to [dec_layers + 1, F*N + T, D] per clip) is opt-in: the reference's only caller discards it (:896) and it costs
        a copy of every row after every layer plus the image rows of the last layer.  Empty unless requested."""
        if not output_hidden_states:
            _, vis = self.forward_image_enc(x)
            logits = self.forward_decoder(y, vis)
            return [l[None] for l in logits], [v[None] for v in vis], []
        self._drain()
        self._call("gitcap_hidden_states_enable", 1)
        try:
            _, vis = self.forward_image_enc(x)
            logits = self.forward_decoder(y, vis)
            B, S_img, T = vis.shape[0], vis.shape[1], logits.shape[1]
            hid = torch.empty((B, self.cfg.dec_layers + 1, S_img + T, self.cfg.dec_width), dtype=torch.float32, device=self._dev)
            with torch.cuda.device(self._dev):
                self._call("gitcap_hidden_states_read", B, S_img, T, ctypes.c_void_p(hid.data_ptr()), self._stream())
        finally:
            self._call("gitcap_hidden_states_enable", 0)
            self._last_memory = None
        return [l[None] for l in logits], [v[None] for v in vis], [h for h in hid]

    @torch.no_grad()
    def greedy_decode(self, src: torch.Tensor, max_len: int = 10, stop: Optional[str] = None) -> torch.Tensor:
        """CLS-prefixed greedy ids [B, 1+steps] (model.py:156-187).  stop='all_sep' is the
        reference rule (break when every row emits SEP in the same step, :184); 'never' always
        runs max_len steps.  The loop runs on the device without per-step host syncs; the
        truncation the reference's `break` implies is applied afterwards."""
        stop = stop or self.stop
        mode = {"all_sep": STOP_ALL_SEP, "never": STOP_NEVER}[stop]
        if max_len > self.max_text_len:
            raise ValueError(f"max_len {max_len} > max_text_len={self.max_text_len} the handle was created for")
        if src.device.type == "cpu":              # CPU tensor in -> CPU ids out (real_time_inference.py:57): the copy back is a
            return self._vouched(lambda: self._greedy_decode(src, max_len, mode))    # synchronisation, so the result is vouched for
        return self._greedy_decode(src, max_len, mode)

    def _greedy_decode(self, src, max_len, mode):
        self._drain()
        fr, raw = self._check_frames(src)         # raw: camera frames [B,F,H,W,3] uint8 BGR, transform fused into the patch gather
        B, F = fr.shape[:2]
        outs, steps_all = [], []
        with torch.cuda.device(self._dev):
            for b0 in range(0, B, self.max_batch):
                chunk = self._to_device(fr[b0:b0 + self.max_batch])      # (a CPU tensor: staged chunk by chunk, stream ordered)
                ids = torch.empty((chunk.shape[0], max_len + 1), dtype=torch.int64, device=self._dev)
                steps = torch.zeros((1,), dtype=torch.int32, device=self._dev)
                if raw:
                    self._call("gitcap_greedy_raw", ctypes.c_void_p(chunk.data_ptr()), chunk.shape[0], F, chunk.shape[2],
                               chunk.shape[3], max_len, mode, ctypes.c_void_p(ids.data_ptr()),
                               ctypes.c_void_p(steps.data_ptr()), self._stream())
                else:
                    self._call("gitcap_greedy", ctypes.c_void_p(chunk.data_ptr()), chunk.shape[0], F, max_len, mode,
                               ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(steps.data_ptr()), self._stream())
                outs.append(ids)
                steps_all.append(steps)
        self._last_memory = None
        ids = torch.cat(outs, 0)
        if mode == STOP_ALL_SEP:
            if len(outs) == 1:
                n = int(steps_all[0].item())
            else:   # the rule is over the WHOLE batch: first step at which every row emitted SEP
                all_sep = (ids[:, 1:] == self.sep_token_id).all(dim=0)
                nz = torch.nonzero(all_sep)
                n = int(nz[0].item()) + 1 if nz.numel() else max_len
            ids = ids[:, :1 + n]
        return ids.to(src.device) if src.device != ids.device else ids

    generate = greedy_decode      # the name BASELINE.json's north_star uses for this entry point

    @torch.no_grad()
    def greedy_decode_async(self, src: torch.Tensor, max_len: int = 10, stop: Optional[str] = None,
                            coalesce: int = 1) -> "CaptionFuture":
        """Pipelined greedy_decode for a stream of batches: returns immediately with a future; up to
        FOUR submissions may be in flight, so one batch's image pass (MFMA bound) overlaps the token
        loops (latency bound) of the batches before it on the handle's internal HIP streams.  Call
        ``.result()`` (in submission order) to make the current stream wait and get the ids.

        `src` may be transformed fp32 frames [B,F,3,S,S] or uint8 BGR camera frames [B,F,H,W,3] (the transform of
        dataloader.py:18-32 then runs on the device, fused with the patch gather), on the device or -- what the reference's
        callers hold -- in HOST memory: a CPU tensor goes through a pinned staging ring (_StagingRing), so its host -> device
        copy runs under the compute of the batches before it (page-locked tensors, e.g. from DataLoader(pin_memory=True), are
        copied from where they are).  A CPU tensor must not be modified before ``result()``.

        ``coalesce=k`` (dynamic batching): k consecutive calls with the same shape are concatenated and
        run as ONE pass of k*B clips (the per-clip results are bitwise the same: the kernels are batch
        invariant); a ``result()`` on a batch that is still waiting for partners flushes it.  Needs
        max_batch >= k*B."""
        stop = stop or self.stop
        mode = {"all_sep": STOP_ALL_SEP, "never": STOP_NEVER}[stop]
        if max_len > self.max_text_len:
            raise ValueError(f"max_len {max_len} > max_text_len={self.max_text_len} the handle was created for")
        fr, raw = self._check_frames(src)         # nothing is copied yet: a CPU batch is staged when its group is submitted
        B = fr.shape[0]
        if B * max(1, coalesce) > self.max_batch:
            raise ValueError(f"batch {B} x coalesce {coalesce} > max_batch={self.max_batch}")
        host = fr.device.type == "cpu"
        if not host:
            fr = self._to_device(fr)
        fut = CaptionFuture(self, fr, mode, max_len, src.device)
        key = (tuple(fr.shape), max_len, mode, raw, host)
        if self._pending and self._pending_key != key:
            self._flush_pending()
        self._pending.append(fut)
        self._pending_key = key
        if len(self._pending) >= max(1, coalesce):
            self._flush_pending()
        return fut

    def _stage_group(self, parts, raw):
        """Host-fed submission: the callers' CPU tensors -> one ring entry (pinned staging unless the single tensor is already
        page-locked) -> device.  Returns (device frames, ring entry, the stream the copy is on = what the submission is ordered behind)."""
        dv, entry, st = self._staging().stage(self, parts, torch.uint8 if raw else torch.float32)
        return dv, entry, ctypes.c_void_p(st.cuda_stream)

    def _flush_pending(self):
        """Submit the waiting batches as one pass and hand each future its row range."""
        group, self._pending = self._pending, []
        if not group:
            return
        raw, host = self._pending_key[3], self._pending_key[4]
        max_len, mode = group[0]._max_len, group[0]._mode
        while len(self._inflight) >= 4:        # four slots: the oldest submission's slot is about to be reused
            self._wait_submission(self._inflight[0])
        with torch.cuda.device(self._dev):
            parts = entry = None
            if host:
                parts = [f._frames for f in group]
                frames, entry, stream = self._stage_group(parts, raw)
            else:
                frames = group[0]._frames if len(group) == 1 else torch.cat([f._frames for f in group], 0)
                stream = self._stream()
            B, F = frames.shape[:2]
            ids = torch.empty((B, max_len + 1), dtype=torch.int64, device=self._dev)
            steps = torch.zeros((1,), dtype=torch.int32, device=self._dev)
            ticket = ctypes.c_int(-1)
            stop = STOP_NEVER if len(group) > 1 else mode       # the stop rule is per caller batch: applied in result()
            if raw:
                self._submit("gitcap_greedy_raw_submit", ctypes.c_void_p(frames.data_ptr()), B, F, frames.shape[2], frames.shape[3],
                             max_len, stop, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(steps.data_ptr()), stream,
                             ctypes.byref(ticket))
            else:
                self._submit("gitcap_greedy_submit", ctypes.c_void_p(frames.data_ptr()), B, F, max_len, stop,
                             ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(steps.data_ptr()), stream, ctypes.byref(ticket))
        self._last_memory = None
        shared = _Submission(ticket.value, frames, (ids, steps), len(group) > 1, users=len(group), parts=parts)
        if entry is not None:
            entry["sub"] = shared
        self._inflight.append(shared)
        r0 = 0
        for f in group:
            n = f._frames.shape[0]
            f._attach(shared, r0, r0 + n)
            r0 += n

    @torch.no_grad()
    def step_logits(self, ids_last: torch.Tensor, t: int, beams: int = 1) -> torch.Tensor:
        """One KV-cached decoding step = `scores = step(input_ids)` of model.py:519: ids_last [rows]
        is the token at text position t of every row; returns fp32 logits [rows, V]."""
        self._drain()
        ids = self._ids(ids_last).view(-1, 1)
        rows = ids.shape[0]
        logits = torch.empty((rows, self.cfg.vocab_size), dtype=torch.float32, device=self._dev)
        with torch.cuda.device(self._dev):
            self._call("gitcap_text_forward", ctypes.c_void_p(ids.data_ptr()), 1, rows, beams, t, 1,
                       ctypes.c_void_p(logits.data_ptr()), 0, None, 0, self._stream())
        return logits

    @torch.no_grad()
    def infer(self, src: torch.Tensor, beam_size: int = 4, max_steps: int = 15, length_penalty: float = 0.6,
              per_node_beam_size: int = 2, num_keep_best: int = 1, save_logits: bool = False,
              on_device: Optional[bool] = None) -> dict:
        """GIT inference with beam search = ``GenerativeImageTextModel.infer`` (model.py:426-462) driven by
        ``GeneratorWithBeamSearchV2.search`` (model.py:479-678; defaults of :702-708).  Returns the
        reference's output dict: predictions [B, max_steps] (CLS-prefixed, EOS padded), logprobs [B,1],
        logits_dict (when save_logits, cf. :521: per-step [B*beams, V] host arrays from the host operator, which stops once every
        clip is done (:640); ONE device tensor [max_steps - 1, B*beams, V] from the device search, which runs every step) and
        visual_features.
        Default: the device-resident search (no host sync per step); on_device=False runs the host-side
        operator of gitcap/search.py (needed for num_keep_best > 1 or save_logits)."""
        from .search import GeneratorWithBeamSearch
        if beam_size > self.max_beams:
            raise ValueError(f"beam_size {beam_size} > max_beams={self.max_beams} the handle was created for")
        if max_steps > self.max_text_len:
            raise ValueError(f"max_steps {max_steps} > max_text_len={self.max_text_len}")
        self._drain()
        fr, raw = self._check_frames(src)
        B, F = fr.shape[:2]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch={self.max_batch}")
        fr = self._to_device(fr)
        if on_device is None:
            on_device = (num_keep_best == 1 and not save_logits and beam_size * per_node_beam_size <= 16
                         and per_node_beam_size >= 2)
        if on_device:
            self._check_device_search(beam_size, per_node_beam_size, num_keep_best)
            decoded, logprobs, steps, vis = self._infer_device(fr, beam_size=beam_size, max_steps=max_steps, length_penalty=length_penalty,
                                                               per_node_beam_size=per_node_beam_size, sync=True,
                                                               save_logits=save_logits, want_visual=False, raw=raw)
            return {"predictions": decoded, "logprobs": logprobs[:, None], "logits_dict": [] if steps is None else steps,
                    "visual_features": None}
        _, vis = self.forward_image_enc(fr)
        searcher = GeneratorWithBeamSearch(self.sep_token_id, max_steps, beam_size, per_node_beam_size, length_penalty)
        start = torch.full((B, 1), self.cls_token_id, dtype=torch.long, device=self._dev)      # model.py:429-431

        def step(ids):                       # decoding_step bound at model.py:442-445, KV-cached
            return self.step_logits(ids[:, -1], ids.shape[1] - 1, beams=beam_size)

        def reorder(beam_idx, cur_len):      # what model.py:623-634 leaves commented out
            self.reorder_rows(beam_idx, cur_len)

        decoded, logprobs, saved = searcher.search(start, step, num_keep_best=num_keep_best, reorder=reorder,
                                                   save_logits=save_logits)
        return {"predictions": decoded, "logprobs": logprobs, "logits_dict": saved, "visual_features": vis}

    def _check_device_search(self, beam_size, per_node_beam_size, num_keep_best):
        if per_node_beam_size < 2:
            raise ValueError("the device-resident search needs per_node_beam_size >= 2: with one candidate per beam "
                             "a single EOS leaves fewer than beam_size live beams (model.py:606 asserts against it); "
                             "the host operator (on_device=False) raises when that happens")
        if num_keep_best != 1:
            raise ValueError("the device-resident search keeps one hypothesis")
        if beam_size * per_node_beam_size > 16:
            raise ValueError("the device-resident search ranks at most 16 candidates per clip")

    def _infer_device(self, fr, *, beam_size, max_steps, length_penalty, per_node_beam_size, sync, save_logits, want_visual,
                      raw=None, stream=None):
        """The device-resident search on frames already on the device (fp32 NCHW, or raw uint8 HWC): synchronously on the current
        stream (sync=True; per-step logits are not available there) or as a pipelined submission ordered behind `stream` (default:
        the current stream; a host-fed submission passes the copy stream).  Returns (decoded, logprobs, step logits | None,
        visual | None) [+ the ticket when submitted]."""
        if raw is None:
            raw = fr.dtype == torch.uint8
        B, F = fr.shape[:2]
        decoded = torch.empty((B, max_steps), dtype=torch.int64, device=self._dev)
        logprobs = torch.empty((B,), dtype=torch.float32, device=self._dev)
        null = ctypes.c_void_p(None)
        with torch.cuda.device(self._dev):
            if sync and not save_logits and not want_visual and not raw:
                self._drain()
                self._call("gitcap_beam_search", ctypes.c_void_p(fr.data_ptr()), B, F, beam_size, max_steps,
                           ctypes.c_float(length_penalty), per_node_beam_size, ctypes.c_void_p(decoded.data_ptr()),
                           ctypes.c_void_p(logprobs.data_ptr()), self._stream())
                self._last_memory = None
                return decoded, logprobs, None, None
            steps = torch.empty((max_steps - 1, B * beam_size, self.cfg.vocab_size), dtype=torch.float32, device=self._dev) if save_logits else None
            vis = torch.empty((B, F * self.cfg.tokens_per_frame, self.cfg.enc_width), dtype=torch.float32, device=self._dev) if want_visual else None
            while len(self._inflight) >= 4:
                self._wait_submission(self._inflight[0])
            ticket = ctypes.c_int(-1)
            tail = (ctypes.c_void_p(vis.data_ptr()) if want_visual else null, beam_size, max_steps, ctypes.c_float(length_penalty),
                    per_node_beam_size, ctypes.c_void_p(decoded.data_ptr()), ctypes.c_void_p(logprobs.data_ptr()),
                    ctypes.c_void_p(steps.data_ptr()) if save_logits else null, stream if stream is not None else self._stream(),
                    ctypes.byref(ticket))
            if raw:
                self._submit("gitcap_beam_search_raw_submit", ctypes.c_void_p(fr.data_ptr()), B, F, fr.shape[2], fr.shape[3], *tail)
            else:
                self._submit("gitcap_beam_search_submit", ctypes.c_void_p(fr.data_ptr()), B, F, *tail)
            self._last_memory = None
            if sync:                           # (a re-run, or a synchronous call that wants logits / visual features / takes raw frames)
                self._call("gitcap_beam_search_wait", ticket.value, self._stream())
                return decoded, logprobs, steps, vis
        return decoded, logprobs, steps, vis, ticket.value

    @torch.no_grad()
    def infer_async(self, src: torch.Tensor, beam_size: int = 4, max_steps: int = 15, length_penalty: float = 0.6,
                    per_node_beam_size: int = 2, save_logits: bool = False, visual_features: bool = False) -> "InferFuture":
        """Pipelined ``infer`` (the device-resident search) for a stream of batches: returns at once with a future; up to FOUR
        submissions (of this kind or of greedy_decode_async) may be in flight, so one batch's image pass overlaps the search loops
        of the batches before it.  ``result()`` returns ``infer``'s dict; with ``save_logits`` its ``logits_dict`` is one device
        tensor [max_steps - 1, B * beam_size, V] (the raw logits of every step, model.py:521), with ``visual_features`` the fp32
        features [B, F*N, Dv] (model.py:460).  Results are bitwise those of the synchronous call.  `src` as in
        greedy_decode_async: fp32 frames or uint8 camera frames, on the device or in host memory (staged through the pinned ring)."""
        self._check_device_search(beam_size, per_node_beam_size, 1)
        if beam_size > self.max_beams:
            raise ValueError(f"beam_size {beam_size} > max_beams={self.max_beams} the handle was created for")
        if max_steps > self.max_text_len:
            raise ValueError(f"max_steps {max_steps} > max_text_len={self.max_text_len}")
        if self._pending:
            self._flush_pending()
        fr, raw = self._check_frames(src)
        if fr.shape[0] > self.max_batch:
            raise ValueError(f"batch {fr.shape[0]} > max_batch={self.max_batch}")
        kw = dict(beam_size=beam_size, max_steps=max_steps, length_penalty=length_penalty, per_node_beam_size=per_node_beam_size)
        parts = entry = stream = None
        if fr.device.type == "cpu":
            while len(self._inflight) >= 4:      # (before the staging: the entry about to be reused belongs to the oldest)
                self._wait_submission(self._inflight[0])
            parts = [fr]
            with torch.cuda.device(self._dev):
                fr, entry, stream = self._stage_group(parts, raw)
        else:
            fr = self._to_device(fr)
        decoded, logprobs, steps, vis, ticket = self._infer_device(fr, sync=False, save_logits=save_logits, want_visual=visual_features,
                                                                   raw=raw, stream=stream, **kw)
        sub = _Submission(ticket, fr, (decoded, logprobs, steps, vis), False, parts=parts)
        if entry is not None:
            entry["sub"] = sub
        self._inflight.append(sub)
        return InferFuture(self, sub, kw, src.device, save_logits)

    def beam_search(self, src: torch.Tensor, max_len: int = 10, k: int = 3) -> torch.Tensor:
        """Signature of StudentCandidateV1.beam_search (model.py:189): best sequence per clip
        [B, max_len], found with the GIT search operator (length penalty 0.6, model.py:702-708)."""
        out = self.infer(src, beam_size=k, max_steps=max_len)["predictions"]
        return out.to(src.device) if src.device != out.device else out

    def reorder_rows(self, src_rows: torch.Tensor, t_len: int):
        """Beam reorder of the text K/V cache (what model.py:623-634 sketches in comments)."""
        idx = src_rows.to(device=self._dev, dtype=torch.int32).contiguous()
        with torch.cuda.device(self._dev):
            self._call("gitcap_reorder_rows", ctypes.c_void_p(idx.data_ptr()), idx.numel(), t_len, self._stream())

    PROF_CLASSES = ("gemm", "attn_full", "skinny", "attn_text", "rowops", "gemm_ln")

    def profile(self, enable: bool):
        self._call("gitcap_profile_enable", int(bool(enable)))

    def profile_read(self) -> dict:
        """{class: {ms, launches, flops, bytes}} summed over the bracketed launches since the last read."""
        out = {}
        for i, name in enumerate(self.PROF_CLASSES):
            ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            n = ctypes.c_int64()
            self._call("gitcap_profile_read", i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
            out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out

    def set_fp8_scale(self, scale: float):
        """Static power-of-two scale of the e4m3 activation codes of compute='fp8_ffn' (gitcap_set_fp8_scale): codes hold
        value / scale and reach +-448.  Synchronises the device."""
        self._drain()
        self._call("gitcap_set_fp8_scale", ctypes.c_float(float(scale)))
        self.fp8_scale = float(scale)
        self._kw["fp8_scale"] = self.fp8_scale
        self._last_memory = None

    def fp8_saturations(self, reset: bool = True) -> int:
        """Activation codes of valid image rows that compute='fp8_ffn' clamped at +-448 since the last reset
        (gitcap_fp8_saturations).  0 for any other compute mode.  Synchronises the device."""
        n = ctypes.c_int64()
        self._call("gitcap_fp8_saturations", ctypes.byref(n), int(bool(reset)))
        return n.value

    def weight_bytes(self) -> int:
        """Device bytes of the loaded tensors (GEMM weights + scales, tables, biases)."""
        n = ctypes.c_int64()
        self._lib.gitcap_weight_bytes(self._handle, ctypes.byref(n))
        return n.value

    def workspace_bytes(self) -> int:
        n = ctypes.c_int64()
        self._lib.gitcap_workspace_bytes(self._handle, ctypes.byref(n))
        return n.value
