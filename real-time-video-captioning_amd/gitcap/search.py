"""Search operator of the caption path: beam search over a ``step`` callable.

Mirrors the interface of ``GeneratorWithBeamSearchV2`` in the reference
(/root/reference/src/models/model.py:465-678; constructed at :702-708 with beam_size 4,
max_steps 15, length_penalty 0.6):

    searcher = GeneratorWithBeamSearch(eos_index, max_steps, beam_size, per_node_beam_size=2,
                                       length_penalty=1.0)
    decoded, logprobs, saved_logits = searcher.search(input_ids, step, num_keep_best=1)

``step(input_ids[B*beams, cur_len]) -> logits[B*beams, V]`` (model.py:519).  What differs from the
reference is WHERE things run: log-softmax + beam-score add + top-(per_node*beams) over beams*V is one
HIP kernel (gitcap_beam_topk) and only its [B, per_node*beams] result crosses to the host once per
step (the reference copies the full [B*beams, V] logits to the host every step, :521, and calls
.item() per candidate, :576-594).  The hypothesis bookkeeping (:573-611, :653-678) is host logic in
the reference and stays host logic here.  The repetition penalty (:522-531) and the sampling branch
(:532-554: temperature, top-k / top-p filtering, multinomial draw) are torch tensor operations on the
device; the reference's teacher never enables them (model.py:702-708, :768), so they are off the hot path.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional

import torch

from . import _lib


class _Hyps:
    """n-best finished hypotheses of one clip (the role of upstream BeamHypotheses, model.py:503)."""

    def __init__(self, n_hyp: int, max_length: int, length_penalty: float):
        self.n_hyp, self.max_length, self.length_penalty = n_hyp, max_length - 1, length_penalty
        self.items: List[tuple] = []          # (score, ids list)
        self.worst = 1e9

    def add(self, ids: List[int], sum_logprobs: float) -> None:
        score = sum_logprobs / len(ids) ** self.length_penalty
        if len(self.items) < self.n_hyp or score > self.worst:
            self.items.append((score, ids))
            if len(self.items) > self.n_hyp:
                order = sorted(range(len(self.items)), key=lambda i: self.items[i][0])
                del self.items[order[0]]
                self.worst = min(s for s, _ in self.items)
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs: float) -> bool:
        if len(self.items) < self.n_hyp:
            return False
        return self.worst >= best_sum_logprobs / self.max_length ** self.length_penalty


class GeneratorWithBeamSearch:
    def __init__(self, eos_index: int, max_steps: int, beam_size: int, per_node_beam_size: int = 2,
                 length_penalty: float = 1.0, repetition_penalty: float = 1.0, temperature: float = 1.0):
        if beam_size * per_node_beam_size > 16:
            raise ValueError("beam_size * per_node_beam_size must be <= 16")
        self._eos_index, self.max_steps, self.beam_size = eos_index, max_steps, beam_size
        self.per_node_beam_size, self.length_penalty = per_node_beam_size, length_penalty
        self.repetition_penalty, self.temperature = float(repetition_penalty), float(temperature)
        self._lib = None

    def _topk(self, logits: torch.Tensor, beam_scores: torch.Tensor, B: int):
        if self._lib is None:
            self._lib = _lib.load()
        beams, V = self.beam_size, logits.shape[-1]
        K = self.per_node_beam_size * beams
        logits = logits.float().contiguous()
        out_s = torch.empty((B, K), dtype=torch.float32, device=logits.device)
        out_i = torch.empty((B, K), dtype=torch.int32, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = self._lib.gitcap_beam_topk(ctypes.c_void_p(logits.data_ptr()), logits.stride(0),
                                            ctypes.c_void_p(beam_scores.data_ptr()), B, beams, V, K,
                                            ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream))
        if rc != 0:
            raise _lib.GitcapError(f"gitcap_beam_topk failed (status {rc})")
        return out_s.cpu(), out_i.cpu()          # the one host sync of the step

    @staticmethod
    def _filter(logits: torch.Tensor, top_k: int, top_p: float, min_tokens_to_keep: int = 2) -> torch.Tensor:
        """top_k_top_p_filtering of model.py:537 (published HF algorithm; see oracle/search_oracle.py)."""
        neg = float("-inf")
        if top_k and top_k > 0:
            kth = torch.topk(logits, min(max(top_k, min_tokens_to_keep), logits.size(-1)))[0][..., -1, None]
            logits = logits.masked_fill(logits < kth, neg)
        if top_p is not None and top_p < 1.0:
            sorted_logits, sorted_idx = torch.sort(logits, descending=True)
            remove = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1) > top_p
            remove[..., :min_tokens_to_keep] = False
            remove = torch.cat([torch.zeros_like(remove[..., :1]), remove[..., :-1]], dim=-1)   # keep the token that crosses top_p
            logits = logits.masked_fill(torch.zeros_like(remove).scatter(-1, sorted_idx, remove), neg)
        return logits

    def _sample(self, scores: torch.Tensor, beam_scores: torch.Tensor, B: int, top_k, top_p, generator):
        """model.py:532-554.  A CPU ``generator`` draws on the host (reproducible against the oracle)."""
        nb, pn, V = self.beam_size, self.per_node_beam_size, scores.shape[-1]
        if self.temperature != 1.0:
            scores = scores / self.temperature
        scores = self._filter(scores, top_k or 0, 1.0 if top_p is None else top_p)
        probs = torch.softmax(scores, dim=-1)
        if generator is not None and generator.device.type == "cpu":
            words = torch.multinomial(probs.cpu(), num_samples=pn, generator=generator).to(scores.device)
        else:
            words = torch.multinomial(probs, num_samples=pn, generator=generator)
        sc = torch.gather(torch.log_softmax(scores, dim=-1), -1, words) + beam_scores[:, None]
        # model.py:549-552 as written: beam offsets tiled over the row, samples beam-major
        offs = (torch.arange(nb, device=scores.device) * V).repeat(B, pn)
        return sc.view(B, pn * nb).cpu(), (words.view(B, pn * nb) + offs).cpu()

    @torch.no_grad()
    def search(self, input_ids: torch.Tensor, step: Callable[[torch.Tensor], torch.Tensor], num_keep_best: int = 1,
               do_sample: bool = False, top_k=None, top_p=None, num_return_sequences: int = 1,
               reorder: Optional[Callable[[torch.Tensor, int], None]] = None, save_logits: bool = False,
               generator: Optional[torch.Generator] = None):
        """Returns (decoded [B, max_steps] EOS padded, logprobs [B, num_keep_best], saved_logits).
        ``reorder(beam_idx, cur_len)`` is called after every step so a KV-cached ``step`` can permute
        its rows (the reference sketches this in comments, model.py:623-634)."""
        if num_return_sequences != 1:
            input_ids = input_ids[:, None, :].expand(input_ids.shape[0], num_return_sequences, input_ids.shape[1])
            input_ids = input_ids.reshape(-1, input_ids.shape[-1])
        dev = input_ids.device
        B, cur_len = input_ids.shape
        nb, eos, max_length = self.beam_size, self._eos_index, self.max_steps
        ids = input_ids.unsqueeze(1).expand(B, nb, cur_len).contiguous().view(B * nb, cur_len)
        hyps = [_Hyps(num_keep_best, max_length, self.length_penalty) for _ in range(B)]
        beam_scores = torch.zeros((B, nb), dtype=torch.float32)
        beam_scores[:, 1:] = -1e9
        beam_scores = beam_scores.view(-1)
        done = [False] * B
        saved = []
        ids_host = ids.cpu()
        while cur_len < max_length:
            logits = step(ids)
            V = logits.shape[-1]
            if save_logits:
                saved.append(logits.detach().float().cpu().numpy())               # model.py:521
            if self.repetition_penalty != 1.0:                                      # model.py:522-531
                logits = logits.float().clone()
                prev = torch.gather(logits, 1, ids)
                logits.scatter_(1, ids, torch.where(prev < 0, prev * self.repetition_penalty, prev / self.repetition_penalty))
            if do_sample:
                next_scores, next_words = self._sample(logits.float(), beam_scores.to(dev), B, top_k, top_p, generator)
            else:
                next_scores, next_words = self._topk(logits, beam_scores.to(dev), B)
            new_scores, new_words, new_src = [], [], []
            for b in range(B):
                done[b] = done[b] or hyps[b].is_done(float(next_scores[b].max()))
                if done[b]:
                    new_scores += [0.0] * nb; new_words += [eos] * nb; new_src += [0] * nb
                    continue
                kept = 0
                for idx, score in zip(next_words[b].tolist(), next_scores[b].tolist()):
                    beam_id, word_id = divmod(idx, V)
                    if word_id == eos or cur_len + 1 == max_length:
                        hyps[b].add(ids_host[b * nb + beam_id, :cur_len].tolist(), score)
                    else:
                        new_scores.append(score); new_words.append(word_id); new_src.append(b * nb + beam_id)
                        kept += 1
                    if kept == nb:
                        break
                if kept == 0:
                    new_scores += [0.0] * nb; new_words += [eos] * nb; new_src += [0] * nb
                elif kept != nb:
                    if not do_sample:
                        raise RuntimeError("beam underflow: fewer than beam_size live candidates (per_node_beam_size too small)")
                    new_scores += [0.0] * (nb - kept); new_words += [eos] * (nb - kept); new_src += [0] * (nb - kept)
            beam_scores = torch.tensor(new_scores, dtype=torch.float32)
            beam_idx = torch.tensor(new_src, dtype=torch.long)
            words = torch.tensor(new_words, dtype=torch.long)
            ids_host = torch.cat([ids_host[beam_idx], words[:, None]], dim=1)
            ids = ids_host.to(dev)
            if reorder is not None:
                reorder(beam_idx, cur_len)
            cur_len += 1
            if all(done):
                break
        decoded = torch.full((B, num_keep_best, max_length), eos, dtype=torch.long)
        logprobs = torch.full((B, num_keep_best), -1e5)
        for b, h in enumerate(hyps):
            best = sorted(h.items, key=lambda x: -x[0])[:num_keep_best]
            for j, (conf, seq) in enumerate(best):
                decoded[b, j, :len(seq)] = torch.tensor(seq, dtype=torch.long)
                decoded[b, j, len(seq)] = eos
                logprobs[b, j] = conf
        if num_keep_best == 1:
            decoded = decoded.squeeze(1)
        return decoded.to(dev), logprobs.to(dev), saved
