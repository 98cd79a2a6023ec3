"""Offline WordPiece detokeniser for caption ids.

The reference decodes with ``BertTokenizer.from_pretrained('bert-base-uncased')``
(/root/reference/src/models/model.py:733, src/real_time_inference.py:35, :59), a hub download that is
not available offline.  Given a local BERT ``vocab.txt`` (one token per line, line number = id) this
class reproduces ``tokenizer.decode(ids, skip_special_tokens=True)`` for uncased WordPiece: special
tokens ([PAD] [UNK] [CLS] [SEP] [MASK]) are dropped, ``##`` continuation pieces are glued to the previous
piece, and the clean-up of spaces before punctuation follows transformers'
``clean_up_tokenization`` rules.  It has a ``decode`` method with the call shape the reference uses, so
an instance can be passed as ``GitCaptioner(tokenizer=...)``.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence


class WordPieceDecoder:
    SPECIAL = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")

    def __init__(self, vocab: Sequence[str] | str):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = [line.rstrip("\n") for line in f]
        self.vocab: List[str] = list(vocab)
        self.ids = {t: i for i, t in enumerate(self.vocab)}
        self.cls_token_id = self.ids.get("[CLS]")
        self.sep_token_id = self.ids.get("[SEP]")
        self.pad_token_id = self.ids.get("[PAD]")

    def convert_ids_to_tokens(self, ids: Iterable[int]) -> List[str]:
        return [self.vocab[i] if 0 <= int(i) < len(self.vocab) else "[UNK]" for i in ids]

    def decode(self, ids: Iterable[int], skip_special_tokens: bool = True) -> str:
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        out: List[str] = []
        for tok in self.convert_ids_to_tokens(ids):
            if skip_special_tokens and tok in self.SPECIAL:
                continue
            if tok.startswith("##") and out:
                out[-1] += tok[2:]
            else:
                out.append(tok)
        text = " ".join(out)
        for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"),
                     (" 's", "'s"), (" 've", "'ve"), (" 're", "'re")):
            text = text.replace(a, b)
        return text

    def batch_decode(self, batch, skip_special_tokens: bool = True) -> List[str]:
        return [self.decode(row, skip_special_tokens) for row in batch]
