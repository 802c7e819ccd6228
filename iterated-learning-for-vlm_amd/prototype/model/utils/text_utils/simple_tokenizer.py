"""Byte-level BPE tokenizer of OpenAI CLIP with the reference's extra <|mask|> token (vocabulary 49409:
256 byte symbols, 256 end-of-word byte symbols, 48894 merges, <|mask|>=49406, <|startoftext|>=49407,
<|endoftext|>=49408) -- same ids as reference prototype/model/utils/text_utils/simple_tokenizer.py:63-135.
Own implementation: merge ranks in a dict, words merged on a symbol list; the vocabulary file
(bpe_simple_vocab_16e6.txt.gz) is supplied by the user (config key model.kwargs.text_encode.bpe_path)."""
import gzip
import html
import os

import regex as re

try:                                   # the reference cleans text with ftfy when present; identity on clean text
    from ftfy import fix_text as _fix_text
except Exception:                      # pragma: no cover
    def _fix_text(t):
        return t

N_MERGES = 49152 - 256 - 2
_SPECIALS = ["<|mask|>", "<|startoftext|>", "<|endoftext|>"]
_WORD_RE = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                      re.IGNORECASE)


def byte_symbols():
    """byte value -> printable unicode character (GPT-2 convention); insertion order defines ids 0..255"""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table = {b: chr(b) for b in keep}
    extra = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer(object):
    def __init__(self, bpe_path=None):
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("BPE vocabulary %r not found: pass text_encode.bpe_path (bpe_simple_vocab_16e6.txt.gz) "
                                    "or feed pre-tokenised (tokens, pad_mask) pairs to the model" % (bpe_path,))
        self.byte_encoder = byte_symbols()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        self._vocab_text = gzip.open(bpe_path).read()
        lines = self._vocab_text.decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        symbols = list(self.byte_encoder.values())
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + _SPECIALS
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self._cache = {}

    def _merge_word(self, token):
        """greedy lowest-rank pair merging of one pre-token; returns the list of BPE symbols"""
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        if token in _SPECIALS:
            out = [token]
        else:
            word = list(token[:-1]) + [token[-1] + "</w>"]
            while len(word) > 1:
                best, best_rank = None, None
                for pair in zip(word[:-1], word[1:]):
                    r = self.bpe_ranks.get(pair)
                    if r is not None and (best_rank is None or r < best_rank):
                        best, best_rank = pair, r
                if best is None:
                    break
                merged, i = [], 0
                while i < len(word):
                    if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                        merged.append(best[0] + best[1])
                        i += 2
                    else:
                        merged.append(word[i])
                        i += 1
                word = merged
            out = word
        self._cache[token] = out
        return out

    def encode(self, text):
        text = html.unescape(html.unescape(_fix_text(text))).strip()
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for piece in _WORD_RE.findall(text):
            mapped = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge_word(mapped))
        return ids

    def vocab_text(self):
        """the decompressed vocabulary file (what ilvlm_tokenizer_create takes)"""
        return self._vocab_text

    def decode(self, tokens):
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


class NativeTokenizer(object):
    """Batch tokenisation through the C++ BPE of libilvlm_hip.so (ilvlm_tokenizer_*, csrc/tokenizer.cpp): printable-ASCII
    captions are encoded natively; the rest (non-ASCII, HTML entities) are returned by the library as `fallback` rows and
    go through SimpleTokenizer above.  Output layout is TextTransformer.tokenize's (text_transformer.py:155-202)."""

    def __init__(self, simple):
        import ctypes as C
        from ..... import lib as L
        self._C, self._L, self.simple = C, L, simple
        self._h = C.c_void_p()
        text = simple.vocab_text()
        L.check(L.load().ilvlm_tokenizer_create(text, len(text), C.byref(self._h)), "tokenizer_create")

    def __del__(self):
        try:
            if self._h:
                self._L.load().ilvlm_tokenizer_destroy(self._h)
                self._h = None
        except Exception:        # interpreter shutdown
            pass

    def encode_batch(self, texts, ctx, sot, eot):
        """-> (tokens int64 [n, ctx], pad_mask fp32 [n, ctx], lengths int64 [n]) as torch CPU tensors"""
        import numpy as np
        import torch
        C = self._C
        n = len(texts)
        tokens = np.zeros((n, ctx), dtype=np.int64)
        mask = np.full((n, ctx), -np.inf, dtype=np.float32)
        lengths = np.ones(n, dtype=np.int32)
        fallback = np.zeros(n, dtype=np.uint8)
        # embedded NULs cannot cross a C string: such captions take the Python path
        raw = [t.encode("utf-8") for t in texts]
        nul = [i for i, r in enumerate(raw) if b"\0" in r]
        for i in nul:
            raw[i] = b"\x7f"
        arr = (C.c_char_p * n)(*raw)
        self._L.check(self._L.load().ilvlm_tokenizer_encode(self._h, arr, n, ctx, tokens.ctypes.data, mask.ctypes.data,
                                                            lengths.ctypes.data, fallback.ctypes.data), "tokenizer_encode")
        for i in np.nonzero(fallback)[0]:
            toks = [sot] + self.simple.encode(texts[i]) + [eot]
            if len(toks) > ctx:
                toks = [toks[0]] + toks[1:ctx - 1] + [toks[-1]]
            tokens[i, :len(toks)] = toks
            mask[i, :len(toks)] = 0
            lengths[i] = len(toks)
        return torch.from_numpy(tokens), torch.from_numpy(mask), torch.from_numpy(lengths.astype(np.int64))
