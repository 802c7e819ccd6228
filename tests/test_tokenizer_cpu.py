"""C++ BPE tokenizer (ilvlm_tokenizer_*, host only -- runs without a GPU) against the reference's ids (G9 fixture made by
importing the reference, tests/golden/make_golden.py g9) and against the package's own full-Unicode Python tokenizer."""
import ctypes as C
import json
import os
import random
import string

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def toks(golden_dir):
    from ilvlm_amd.prototype.model.utils.text_utils.simple_tokenizer import SimpleTokenizer, NativeTokenizer
    simple = SimpleTokenizer(os.path.join(golden_dir, "bpe_simple_vocab_16e6.txt.gz"))
    return simple, NativeTokenizer(simple)


def _native_only(native, texts, ctx=77):
    """raw C-ABI call: rows the library handled itself + the fallback flags"""
    import ilvlm_amd.lib as L
    n = len(texts)
    tokens = np.full((n, ctx), -1, dtype=np.int64)
    mask = np.full((n, ctx), 7.0, dtype=np.float32)
    lengths = np.zeros(n, dtype=np.int32)
    fb = np.zeros(n, dtype=np.uint8)
    arr = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
    L.check(L.load().ilvlm_tokenizer_encode(native._h, arr, n, ctx, tokens.ctypes.data, mask.ctypes.data,
                                            lengths.ctypes.data, fb.ctypes.data), "tokenizer_encode")
    return tokens, mask, lengths, fb


def _python_rows(simple, texts, ctx=77):
    out = []
    for t in texts:
        ids = [49407] + simple.encode(t) + [49408]
        if len(ids) > ctx:
            ids = [ids[0]] + ids[1:ctx - 1] + [ids[-1]]
        out.append(ids)
    return out


def test_native_matches_reference_fixture(toks, golden_dir):
    simple, native = toks
    with open(os.path.join(golden_dir, "g9_tokenizer.json")) as f:
        g = json.load(f)
    tok, mask, lengths = native.encode_batch(g["captions"], 77, 49407, 49408)
    assert tok.tolist() == g["tokens"]
    assert (mask == 0).int().tolist() == g["pad_mask_valid"]
    assert lengths.tolist() == [sum(r) for r in g["pad_mask_valid"]]
    # every ASCII caption of the fixture without '&' must be handled by the C++ path itself
    tokens, m, ln, fb = _native_only(native, g["captions"])
    for i, cap in enumerate(g["captions"]):
        ascii_ok = all(32 <= ord(ch) < 127 or ch in "\t\n\r\f\v" for ch in cap) and "&" not in cap
        assert bool(fb[i]) == (not ascii_ok), cap
        if ascii_ok:
            assert tokens[i].tolist() == g["tokens"][i]
            assert ln[i] == sum(g["pad_mask_valid"][i])
            assert np.all((m[i] == 0) == np.array(g["pad_mask_valid"][i], dtype=bool))
            assert np.all(np.isneginf(m[i][ln[i]:]))
        else:
            assert np.all(tokens[i] == -1) and np.all(m[i] == 7.0)     # untouched rows


def test_native_matches_python_on_random_ascii(toks):
    simple, native = toks
    rng = random.Random(1234)
    words = ["photo", "a", "the", "Dog's", "isn't", "we'll", "they've", "I'm", "you'd", "it's", "'tis", "rock'n'roll", "x-ray",
             "3.14159", "2024", "<|startoftext|>", "<|endoftext|>", "<|mask|>", "<|", "|>", "e-mail@host.com", "C++", "#1",
             "...", "!!!", "(parenthesised)", "UPPER", "MiXeD", "snake_case", "semi;colon", "tab\tsep", "new\nline", "  ",
             "\r\n", "\f", "\v", "a" * 40, "~`^", "\"quoted\"", "'", "''", "'s", "'S", "'RE", "o'clock", "100%", "$9.99", "[x]",
             "{y}", "\\back", "/fwd", "q?", "w!", "z,", "=+*", "@", "_"]
    alphabet = string.ascii_letters + string.digits + string.punctuation.replace("&", "") + "    \t\n"
    texts = []
    for _ in range(600):
        if rng.random() < 0.7:
            texts.append(" ".join(rng.choice(words) for _ in range(rng.randint(0, 30))))
        else:
            texts.append("".join(rng.choice(alphabet) for _ in range(rng.randint(0, 200))))
    texts += ["", " ", "\n\t ", "a", "'", "<|startoftext|><|endoftext|>", "x" * 500, "9" * 100, "!?" * 120]
    tokens, mask, lengths, fb = _native_only(native, texts)
    assert not fb.any()
    ref = _python_rows(simple, texts)
    for i, ids in enumerate(ref):
        assert lengths[i] == len(ids), repr(texts[i])
        assert tokens[i, :len(ids)].tolist() == ids, repr(texts[i])
        assert np.all(tokens[i, len(ids):] == 0)
        assert np.all(mask[i, :len(ids)] == 0) and np.all(np.isneginf(mask[i, len(ids):]))


def test_fallback_rows_and_context_lengths(toks):
    simple, native = toks
    texts = ["café au lait", "fish &amp; chips", "plain ascii", "中文 caption", "bell\x07char", "del\x7fchar",
             "R&D", "emoji \U0001F600 here"]
    _, _, _, fb = _native_only(native, texts)
    assert fb.tolist() == [1, 1, 0, 1, 1, 1, 1, 1]
    for ctx in (8, 16, 77):
        tok, mask, lengths = native.encode_batch(texts + ["one two three four five six seven eight nine ten"], ctx, 49407, 49408)
        ref = _python_rows(simple, texts + ["one two three four five six seven eight nine ten"], ctx)
        for i, ids in enumerate(ref):
            assert tok[i, :len(ids)].tolist() == ids and lengths[i] == len(ids)
            assert tok[i, 0] == 49407 and tok[i, len(ids) - 1] == 49408
            assert torch.all(mask[i, :len(ids)] == 0) and torch.all(torch.isinf(mask[i, len(ids):]))
    # embedded NUL goes through the Python path too
    tok, _, lengths = native.encode_batch(["nul\0inside"], 77, 49407, 49408)
    assert tok[0, :lengths[0]].tolist() == _python_rows(simple, ["nul\0inside"])[0]


def test_tokenizer_rejects_bad_input(toks):
    import ilvlm_amd.lib as L
    h = C.c_void_p()
    assert L.load().ilvlm_tokenizer_create(b"#version\nnot a merge\n", 21, C.byref(h)) != 0
    assert b"tokenizer_create" in L.load().ilvlm_last_error()
    simple, native = toks
    with pytest.raises(RuntimeError):
        native.encode_batch(["x"], 1, 49407, 49408)          # context_length < 2
