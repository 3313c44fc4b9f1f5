"""Byte-pair tokenisation of prompts for the CLIP text tower (host logic).

The reference calls ``clip.tokenize(text, truncate=True)`` (``mogen/models/transformers/diffusion_transformer.py:145``)
of the un-vendored ``clip`` package (openai/CLIP, ``clip/simple_tokenizer.py``).  Its vocabulary file
(``bpe_simple_vocab_16e6.txt.gz``) ships inside that package, so it is not available offline: the VOCABULARY is
unpinned; the algorithm is pinned against ``transformers.CLIPTokenizer`` on a shared synthetic vocabulary
(tests/test_host.py).  When the
package is importable ``text_encoder.NativeTextEncoder.encode_text`` uses it directly; this module restates the
published scheme for installations that only carry the vocabulary file:

    clean-up (html unescape, whitespace collapse, lower-case) -> regex split into words / digits / punctuation runs ->
    each word's UTF-8 bytes mapped to printable code points, last symbol tagged ``</w>`` -> lowest-rank adjacent pair
    merged repeatedly (ranks = line order of the merges file) -> ids; ``<|startoftext|>`` ... ``<|endoftext|>``,
    zero padding to 77, truncation keeps the end token.

Not restated: the ``ftfy`` mojibake repair the package applies first (identity on well-formed text).
"""
import gzip
import html
from functools import lru_cache

import numpy as np
import regex

SOT, EOT = '<|startoftext|>', '<|endoftext|>'
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)


def byte_symbols():
    """256 byte values -> printable code points: the visible latin-1 ranges map to themselves, the rest to 256, 257, ..."""
    keep = list(range(ord('!'), ord('~') + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipBPE:
    def __init__(self, bpe_path, vocab_size=49408):
        opener = gzip.open if str(bpe_path).endswith('.gz') else open
        with opener(bpe_path, 'rb') as f:
            lines = f.read().decode('utf-8').split('\n')
        n_merges = vocab_size - 256 - 256 - 2
        merges = [tuple(line.split()) for line in lines[1:1 + n_merges] if line.strip()]
        self.byte_sym = byte_symbols()
        # id order: byte symbols in the order the package enumerates them (visible ranges first, then the remapped ones),
        # the same with the end-of-word tag, one entry per merge, the two markers
        keep = [b for b in range(256) if self.byte_sym[b] == chr(b)]
        base = [self.byte_sym[b] for b in keep + [b for b in range(256) if b not in keep]]
        vocab = base + [s + '</w>' for s in base] + [''.join(m) for m in merges] + [SOT, EOT]
        self.ids = {s: i for i, s in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.ids[SOT], self.ids[EOT]
        self._word = lru_cache(maxsize=65536)(self._merge_word)

    def _merge_word(self, token):
        sym = list(token[:-1]) + [token[-1] + '</w>']
        while len(sym) > 1:
            best = min(zip(sym[:-1], sym[1:]), key=lambda p: self.rank.get(p, float('inf')))
            if best not in self.rank:
                break
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1])
                    i += 2
                else:
                    out.append(sym[i])
                    i += 1
            sym = out
        return tuple(sym)

    def encode(self, text):
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r'\s+', ' ', text).strip().lower()
        out = []
        for tok in _SPLIT.findall(text):
            if tok in (SOT, EOT):
                out.append(self.ids[tok])
                continue
            mapped = ''.join(self.byte_sym[b] for b in tok.encode('utf-8'))
            out += [self.ids[s] for s in self._word(mapped)]
        return out

    def tokenize(self, texts, context_length=77, truncate=True):
        """-> int64 array [len(texts), context_length] like ``clip.tokenize`` (zeros after the end token)."""
        if isinstance(texts, str):
            texts = [texts]
        res = np.zeros((len(texts), context_length), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f'Input {t} is too long for context length {context_length}')
                ids = ids[:context_length]
                ids[-1] = self.eot
            res[i, :len(ids)] = ids
        return res
