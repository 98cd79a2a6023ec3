"""CPU oracle of the GIT search operator.  TEST INFRASTRUCTURE ONLY (see oracle/git_oracle.py).

Restates, in plain Python/torch, ``GeneratorWithBeamSearchV2.search``
(/root/reference/src/models/model.py:479-678) and the ``BeamHypotheses`` container it instantiates
(:503).  ``BeamHypotheses`` itself lives in the absent ``generativeimage2text`` package (model.py:15);
it is the XLM/HF container whose published algorithm is: keep the ``n_hyp`` best finished
hypotheses ranked by ``sum_logprobs / len(hyp) ** length_penalty``; ``is_done(best_sum_logprobs)``
is False while fewer than ``n_hyp`` are stored, True if ``early_stopping``, otherwise
``worst_score >= best_sum_logprobs / max_length ** length_penalty``.

PARITY: unpinned by the reference (no test or fixture exercises the search, SURVEY.md par. 4); this
restatement is checked against exhaustive enumeration on small problems (tests/test_search.py).
Defaults mirror model.py:702-708 (beam 4, max_steps 15, length_penalty 0.6) and the upstream
constructor default per_node_beam_size = 2.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.nn.functional as F


class BeamHypotheses:
    def __init__(self, n_hyp: int, max_length: int, length_penalty: float, early_stopping: bool):
        self.max_length = max_length - 1          # ignoring bos
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.n_hyp = n_hyp
        self.hyp: List[Tuple[float, torch.Tensor]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.hyp)

    def add(self, hyp: torch.Tensor, sum_logprobs: float):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self) < self.n_hyp or score > self.worst_score:
            self.hyp.append((score, hyp))
            if len(self) > self.n_hyp:
                sorted_scores = sorted([(s, idx) for idx, (s, _) in enumerate(self.hyp)])
                del self.hyp[sorted_scores[0][1]]
                self.worst_score = sorted_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float) -> bool:
        if len(self) < self.n_hyp:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / self.max_length ** self.length_penalty


def top_k_top_p_filtering(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0, filter_value: float = -float("inf"),
                          min_tokens_to_keep: int = 1) -> torch.Tensor:
    """The filter model.py:537 calls.  It lives in the absent ``generativeimage2text`` package (model.py:16),
    which carries the published HF ``top_k_top_p_filtering``: keep the ``top_k`` largest logits (at least
    ``min_tokens_to_keep``); then drop the tail of the descending-sorted distribution whose cumulative
    probability exceeds ``top_p``, always keeping the first token that crosses it and at least
    ``min_tokens_to_keep`` tokens."""
    logits = logits.clone()
    if top_k and top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        logits[logits < torch.topk(logits, top_k)[0][..., -1, None]] = filter_value
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cumulative = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative > top_p
        if min_tokens_to_keep > 1:
            remove[..., :min_tokens_to_keep] = False
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        logits[remove.scatter(-1, sorted_indices, remove)] = filter_value
    return logits


def beam_search(input_ids: torch.Tensor, step: Callable[[torch.Tensor], torch.Tensor], *, eos_index: int,
                max_steps: int = 15, beam_size: int = 4, per_node_beam_size: int = 2,
                length_penalty: float = 0.6, num_keep_best: int = 1, repetition_penalty: float = 1.0,
                temperature: float = 1.0, do_sample: bool = False, top_k=None, top_p=None, generator=None):
    """model.py:479-678: the greedy-beam branch (do_sample=False) and the sampling branch (:532-554), with the
    repetition penalty of :522-531.  ``step(ids[B*beams, cur_len]) -> logits[B*beams, V]`` of the last position
    (model.py:519).  ``generator`` seeds torch.multinomial (the reference uses the global RNG).
    Returns (decoded [B, max_steps] padded with EOS, logprobs [B, num_keep_best], saved_logits)."""
    batch_size, cur_len = input_ids.shape
    num_beams, pad_token_id = beam_size, eos_index
    input_ids = input_ids.unsqueeze(1).expand(batch_size, num_beams, cur_len).contiguous().view(batch_size * num_beams, cur_len)
    max_length = max_steps                                                             # :500
    hyps = [BeamHypotheses(num_keep_best, max_length, length_penalty, early_stopping=False) for _ in range(batch_size)]
    beam_scores = torch.zeros((batch_size, num_beams), dtype=torch.float)
    beam_scores[:, 1:] = -1e9                                                          # :509
    beam_scores = beam_scores.view(-1)
    done = [False] * batch_size
    saved_logits = []
    while cur_len < max_length:                                                        # :518
        scores = step(input_ids)
        vocab = scores.shape[-1]
        saved_logits.append(scores.detach().clone())
        scores = scores.float().clone()
        if repetition_penalty != 1.0:                                                  # :522-531
            for i in range(batch_size * num_beams):
                for previous_token in set(input_ids[i].tolist()):
                    if scores[i, previous_token] < 0:
                        scores[i, previous_token] *= repetition_penalty
                    else:
                        scores[i, previous_token] /= repetition_penalty
        if do_sample:                                                                  # :532-554
            if temperature != 1.0:
                scores = scores / temperature
            scores = top_k_top_p_filtering(scores, top_k=top_k or 0, top_p=1.0 if top_p is None else top_p, min_tokens_to_keep=2)
            next_words = torch.multinomial(F.softmax(scores, dim=-1), num_samples=per_node_beam_size, generator=generator)
            _scores = torch.gather(F.log_softmax(scores, dim=-1), -1, next_words)
            next_scores = _scores + beam_scores[:, None].expand_as(_scores)
            # (:549-552 as written: the beam offsets are TILED over the row while the samples are beam-major)
            beam_indices = (torch.arange(num_beams) * vocab).repeat(batch_size, per_node_beam_size)
            next_words = next_words.view(batch_size, per_node_beam_size * num_beams) + beam_indices
            next_scores = next_scores.view(batch_size, per_node_beam_size * num_beams)
        else:
            scores = F.log_softmax(scores, dim=-1)                                     # :557
            _scores = (scores + beam_scores[:, None]).view(batch_size, num_beams * vocab)  # :561-563
            next_scores, next_words = torch.topk(_scores, per_node_beam_size * num_beams, dim=1, largest=True, sorted=True)
        next_batch_beam = []
        for b in range(batch_size):                                                    # :573
            done[b] = done[b] or hyps[b].is_done(next_scores[b].max().item())
            if done[b]:
                next_batch_beam.extend([(0, pad_token_id, 0)] * num_beams)
                continue
            next_sent_beam = []
            for idx, score in zip(next_words[b], next_scores[b]):
                beam_id, word_id = int(idx) // vocab, int(idx) % vocab
                if word_id == eos_index or cur_len + 1 == max_length:                  # :592
                    hyps[b].add(input_ids[b * num_beams + beam_id, :cur_len].clone(), score.item())
                else:
                    next_sent_beam.append((score, word_id, b * num_beams + beam_id))
                if len(next_sent_beam) == num_beams:
                    break
            if cur_len + 1 == max_length:
                assert len(next_sent_beam) == 0
            elif not do_sample:
                assert len(next_sent_beam) == num_beams
            elif 0 < len(next_sent_beam) < num_beams:      # sampling may draw EOS often: pad like a finished sentence
                next_sent_beam += [(0, pad_token_id, 0)] * (num_beams - len(next_sent_beam))
            if len(next_sent_beam) == 0:
                next_sent_beam = [(0, pad_token_id, 0)] * num_beams
            next_batch_beam.extend(next_sent_beam)
        beam_scores = torch.tensor([float(x[0]) for x in next_batch_beam])
        beam_words = torch.tensor([x[1] for x in next_batch_beam], dtype=torch.long)
        beam_idx = torch.tensor([x[2] for x in next_batch_beam], dtype=torch.long)
        input_ids = torch.cat([input_ids[beam_idx, :], beam_words.unsqueeze(1)], dim=-1)   # :620-621
        cur_len += 1
        if all(done):
            break
    tgt_len = torch.ones(batch_size, num_keep_best, dtype=torch.long)                 # :653
    logprobs = torch.full((batch_size, num_keep_best), -1e5)
    all_best = []
    for i, h in enumerate(hyps):
        best = []
        hyp_scores = torch.tensor([x[0] for x in h.hyp])
        _, best_idx = torch.topk(hyp_scores, min(num_keep_best, len(hyp_scores)), largest=True)
        for bi, hi in enumerate(best_idx):
            conf, best_hyp = h.hyp[hi]
            best.append(best_hyp)
            logprobs[i, bi] = conf
            tgt_len[i, bi] = len(best_hyp) + 1
        all_best.append(best)
    decoded = torch.full((batch_size, num_keep_best, max_length), pad_token_id, dtype=torch.long)
    for b, best in enumerate(all_best):
        for bi, hypo in enumerate(best):
            decoded[b, bi, : tgt_len[b, bi] - 1] = hypo
            decoded[b, bi, tgt_len[b, bi] - 1] = eos_index
    if num_keep_best == 1:
        decoded = decoded.squeeze(1)
    return decoded, logprobs, saved_logits


def teacher_output(predictions: torch.Tensor, logits_dict, cap: str, num_beams: int = 4) -> torch.Tensor:
    """Oracle of the winning-beam logit gather of ``GenerativeImageTextTeacher.forward``
    (/root/reference/src/models/model.py:771-788), for ONE clip:

    * ``predictions`` [1, max_steps] -- the clip's best hypothesis (``result['predictions']``, :771/:780),
    * ``logits_dict`` -- per search step the [num_beams, V] logits of that clip's beams (``result['logits_dict']``, :772/:776),
    * ``cap`` -- the decoded caption (:771); n = min(words in cap, saved steps) (:772).

    For each of the first n predicted words: the beam whose logit AT THAT WORD is highest (:784-785), then that beam's whole
    logit row (:787-788) -> [1, n, V].  The reference hard-codes 4 beams (:782); ``num_beams`` keeps that as the default."""
    import numpy as np
    n = min(len(cap.split(' ')), len(logits_dict))
    dist = torch.from_numpy(np.array([np.asarray(l) for l in logits_dict[:n]]))               # [n, beams, V]   (:776)
    word_tokens = predictions[0, 1:n + 1].cpu()[:, None, None].expand(-1, num_beams, -1)       # [n, beams, 1]   (:780-781)
    idx = torch.gather(dist, dim=2, index=word_tokens).squeeze(-1).argmax(dim=1)              # [n]             (:784-785)
    idx = idx[:, None, None].expand(-1, -1, dist.shape[-1])                                  # (:787)
    return torch.gather(dist, dim=1, index=idx).squeeze(1)[None, ...]                        # [1, n, V]       (:788)
