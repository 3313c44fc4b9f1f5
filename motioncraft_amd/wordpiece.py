"""WordPiece tokenisation for the evaluation text encoder (host logic).

The reference calls ``AutoTokenizer.from_pretrained(modelpath)(texts, return_tensors='pt', padding=True)``
(``mogen/models/rnns/t2m_bigru_smplx.py:229,276``) on a DistilBERT directory.  The vocabulary (``vocab.txt``) ships with
that directory, so the published BERT scheme is restated here and the ``transformers`` package is not needed at run
time: text clean-up -> whitespace split -> lower-casing + accent stripping (uncased models) -> punctuation split ->
greedy longest-match-first word pieces with the ``##`` continuation prefix -> ``[CLS] ... [SEP]`` -> right padding
with ``[PAD]`` and the attention mask.  Pinned against the ``transformers`` tokenizer by tests/golden/make_golden.py.
"""
import json
import os
import unicodedata

import numpy as np


def _is_space(ch):
    return ch in ' \t\n\r' or unicodedata.category(ch) == 'Zs'


def _is_control(ch):
    return ch not in '\t\n\r' and unicodedata.category(ch).startswith('C')


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith('P')


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    def __init__(self, vocab, lower=True, unk='[UNK]', cls='[CLS]', sep='[SEP]', pad='[PAD]', max_chars=100):
        """vocab: path of ``vocab.txt`` (one piece per line, id = line number), a model directory holding it, or a list."""
        if isinstance(vocab, str):
            d = vocab if os.path.isdir(vocab) else os.path.dirname(vocab)
            path = os.path.join(vocab, 'vocab.txt') if os.path.isdir(vocab) else vocab
            cfg = os.path.join(d, 'tokenizer_config.json')
            if os.path.isfile(cfg):
                with open(cfg) as f:
                    lower = json.load(f).get('do_lower_case', lower)
            with open(path, encoding='utf-8') as f:
                vocab = [line.rstrip('\n') for line in f]
        self.ids = {}
        for i, piece in enumerate(vocab):
            self.ids.setdefault(piece, i)
        self.lower, self.max_chars = lower, max_chars
        for name in (unk, cls, sep, pad):
            if name not in self.ids:
                raise ValueError(f'vocabulary has no {name} entry')
        self.unk, self.cls, self.sep, self.pad = (self.ids[n] for n in (unk, cls, sep, pad))

    def words(self, text):
        kept = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                kept.append(f' {ch} ')
            else:
                kept.append(' ' if _is_space(ch) else ch)
        out = []
        for w in unicodedata.normalize('NFC', ''.join(kept)).split():
            if self.lower:
                w = ''.join(c for c in unicodedata.normalize('NFD', w.lower()) if unicodedata.category(c) != 'Mn')
            cur = ''
            for ch in w:
                if _is_punct(ch):
                    if cur:
                        out.append(cur)
                    out.append(ch)
                    cur = ''
                else:
                    cur += ch
            if cur:
                out.append(cur)
        return out

    def pieces(self, word):
        if len(word) > self.max_chars:
            return [self.unk]
        got, start = [], 0
        while start < len(word):
            end = len(word)
            while end > start:
                piece = ('##' if start else '') + word[start:end]
                if piece in self.ids:
                    break
                end -= 1
            if end == start:
                return [self.unk]
            got.append(self.ids[piece])
            start = end
        return got

    def encode(self, text):
        ids = [self.cls]
        for w in self.words(text):
            ids += self.pieces(w)
        return ids + [self.sep]

    def __call__(self, texts):
        """list of strings -> (input_ids int32 [B, S], attention_mask uint8 [B, S]), S = longest sequence in the batch."""
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t) for t in texts]
        S = max(len(r) for r in rows)
        ids = np.full((len(rows), S), self.pad, dtype=np.int32)
        mask = np.zeros((len(rows), S), dtype=np.uint8)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
            mask[i, :len(r)] = 1
        return ids, mask
