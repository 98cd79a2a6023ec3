"""Offline WordPiece detokeniser: against the `tokenizers` WordPiece decoder on a small local vocabulary,
plus the apostrophe clean-up rule of transformers' clean_up_tokenization (what the reference's
BertTokenizer.decode applies, src/real_time_inference.py:59)."""
from gitcap.tokenizer import WordPieceDecoder

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "man", "is", "play", "##ing", "guitar", ".", ",", "the",
         "dog", "run", "##s", "it", "'", "s", "fast", "!", "skate", "##board", "##er"]


def test_decode_matches_tokenizers_library(tmp_path):
    from tokenizers import Tokenizer, decoders, models
    path = tmp_path / "vocab.txt"
    path.write_text("\n".join(VOCAB) + "\n")
    dec = WordPieceDecoder(str(path))
    ref = Tokenizer(models.WordPiece(vocab={t: i for i, t in enumerate(VOCAB)}, unk_token="[UNK]"))
    ref.add_special_tokens(list(WordPieceDecoder.SPECIAL))
    ref.decoder = decoders.WordPiece(prefix="##", cleanup=True)
    cases = [[2, 5, 6, 7, 8, 9, 10, 11, 3, 3, 3], [2, 13, 14, 15, 16, 12, 17, 7, 20, 21, 3],
             [2, 5, 22, 23, 24, 3, 0, 0], [2, 3], [2, 1, 6, 4, 3]]
    for ids in cases:
        assert dec.decode(ids, skip_special_tokens=True) == ref.decode(ids, skip_special_tokens=True), ids
    assert dec.cls_token_id == 2 and dec.sep_token_id == 3 and dec.pad_token_id == 0
    assert dec.batch_decode(cases[:2]) == [ref.decode(c, skip_special_tokens=True) for c in cases[:2]]
    assert dec.decode([2, 999999, 6, 3]) == "man"            # out-of-range id -> [UNK] -> dropped as special
    assert dec.decode([2, 5, 6, 3], skip_special_tokens=False) == "[CLS] a man [SEP]"


def test_apostrophe_cleanup_follows_transformers_rule():
    # transformers.clean_up_tokenization: " ' " -> "'"  ("it ' s" -> "it's")
    dec = WordPieceDecoder(VOCAB)
    assert dec.decode([2, 17, 18, 19, 20, 3]) == "it's fast"
