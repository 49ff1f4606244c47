#!/usr/bin/env python3
"""Generate tests/golden/tokenizer/* — fixtures for tinygpt_amd/host/tokenizer.cpp and regex.cpp.

Runs ONLY in the build container.  What it writes (data only):
  gpt2/, Mistral-7B-v0.3/      the tokenizer.json + tokenizer_config.json data files the reference's own tokenizer tests load
                               (assets/tokenizer/*, test/test_tokenizer.cpp:82-84), copied byte for byte;
  reference_vectors.json       the text -> ids known-answer pairs of test/test_tokenizer.cpp for those two tokenizers
                               (:136-156 gpt2, :210-232 Mistral) and the decode rule each test states;
  llama3_style/, qwen2_style/  small tokenizers TRAINED HERE with the `tokenizers` library on a synthetic corpus, with the
                               pipeline shape of Llama-3 (Split regex + ByteLevel, ignore_merges, BOS template) and of
                               Qwen2 (NFC + Split regex + ByteLevel) — the real files are not in the reference tree;
  hf_vectors.json              `tokenizers` outputs (ids + decoded text) for extra texts on gpt2 and the two trained
                               tokenizers;
  regex_vectors.json           match ranges of the three pre-tokenizer patterns on tricky texts, from the `regex` module.
"""
import json
import os
import re
import shutil
import sys

import regex
from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "tokenizer")
REF = "/root/reference"

GPT2_PAT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
LLAMA3_PAT = r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"""
O200K_PAT = (r"""[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?|"""
             r"""[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?|"""
             r"""\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+""")          # GPT-4o style: sub-categories and marks
DSV3_PAT = (r"""[!"#$%&'()*+,\-./:;<=>?@\[\\\]^_`{|}~][A-Za-z]+|[^\r\n\p{L}\p{P}\p{S}]?[\p{L}\p{M}]+| ?[\p{P}\p{S}]+[\r\n]*|"""
            r"""\s*[\r\n]+|\s+(?!\S)|\s+""")                                                    # DeepSeek-V3 style: \p{P} \p{S} \p{M}
QWEN2_PAT = r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"""

TEXTS = [
    "hello world!", "Hello, World! It's 2024 and I'VE got 1234567 apples.", "   leading and trailing   ",
    "tabs\tand\nnewlines\r\n\r\nmixed  \n  spaces", "hello，你好啊, thanks", " ありがとうございます。 Arigatoo gozaimasu",
    "你好😀🐶", "Aujourd'hui, j'ai bu un café très fort.", "é vs é and Å ngstrom 한글 한",
    "snake_case CamelCase kebab-case 3.14159 1e-9 0xDEADBEEF", "x = y**2 + z[0] // {a: b} <tag attr=\"v\"/> &amp;",
    "don't DON'T we'll WE'LL they're", "a", " ", "  ", "\n", "multiple    spaces     here", "ends with space ",
    "Ünïcödé Ωmega ß straße ǅ ǆ ١٢٣ ४५६ Ⅻ ½", "emoji 👩‍👩‍👧‍👦 family and flags 🇯🇵🇺🇸", "the quick brown fox jumps over the lazy dog " * 8,
    "HTTPServer2Go parseXMLFile iPhone15Pro e\u0301le\u0300ve ÉCOLE École", "€100 + $5 = ¥? © 2024 — “quoted” … «guillemets» x≠y ∑∫√ ₿ №5 ‰",
]


def copy_reference_data():
    for name in ("gpt2", "Mistral-7B-v0.3"):
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        for f in ("tokenizer.json", "tokenizer_config.json"):
            shutil.copyfile(os.path.join(REF, "assets", "tokenizer", name, f), os.path.join(d, f))


def c_unescape(s):
    return s.encode("utf-8").decode("unicode_escape").encode("latin-1").decode("utf-8") if "\\" in s else s


def reference_vectors():
    src = open(os.path.join(REF, "test", "test_tokenizer.cpp"), encoding="utf-8").read()
    out = {}
    for test, name, rule in (("tokenizer_gpt2", "gpt2", "text"), ("tokenizer_mistral_7b", "Mistral-7B-v0.3", "bos+space+text")):
        body = src[src.index(f"TEST(TEST_tokenizer, {test})"):]
        body = body[:body.index("for (auto")]
        pairs = []
        for m in re.finditer(r'\{"((?:[^"\\]|\\.)*)",\s*\{([0-9,\s]+)\}\}', body):
            pairs.append({"text": c_unescape(m.group(1)), "ids": [int(x) for x in m.group(2).replace("\n", " ").split(",") if x.strip()]})
        out[name] = {"decode_rule": rule, "pairs": pairs}
    return out


def corpus():
    words = ("the of and to in is that it was for on are as with his they at be this from have or by one had not but what all were when "
             "we there can an your which their said if do will each about how up out them then she many some so these would other into "
             "has more her two like him see time could no make than first been its who now people my made over did down only way find "
             "use may water long little very after words called just where most know hello world thanks direction right putting café "
             "très fort aujourd'hui naïve straße über 你好 谢谢 世界 ありがとう ございます こんにちは 한글 감사합니다").split()
    lines = []
    for i in range(4000):
        n = 5 + (i * 7) % 11
        line = " ".join(words[(i * 13 + k * 17) % len(words)] for k in range(n))
        if i % 3 == 0: line = line.capitalize() + "."
        if i % 5 == 0: line += f" {i * 37 % 100000}"
        if i % 7 == 0: line = "  " + line + "\n"
        lines.append(line)
    return lines


def train(style):
    tok = Tokenizer(models.BPE(ignore_merges=(style == "llama3")))
    pat = LLAMA3_PAT if style == "llama3" else QWEN2_PAT
    if style == "qwen2":
        tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    specials = ["<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>"] if style == "llama3" else ["<|endoftext|>", "<|im_start|>", "<|im_end|>"]
    trainer = trainers.BpeTrainer(vocab_size=1200, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(corpus(), trainer)
    if style == "llama3":
        bos = tok.token_to_id("<|begin_of_text|>")
        tok.post_processor = processors.Sequence([processors.ByteLevel(trim_offsets=False),
                                                  processors.TemplateProcessing(single="<|begin_of_text|> $A", pair="<|begin_of_text|> $A <|begin_of_text|>:1 $B:1",
                                                                                special_tokens=[("<|begin_of_text|>", bos)])])
        cfg = {"bos_token": "<|begin_of_text|>", "eos_token": "<|end_of_text|>", "model_max_length": 131072}
    else:
        tok.post_processor = processors.ByteLevel(trim_offsets=False)
        cfg = {"add_bos_token": False, "bos_token": None, "eos_token": "<|im_end|>", "pad_token": "<|endoftext|>", "model_max_length": 32768}
    d = os.path.join(OUT, f"{style}_style")
    os.makedirs(d, exist_ok=True)
    tok.save(os.path.join(d, "tokenizer.json"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    return Tokenizer.from_file(os.path.join(d, "tokenizer.json"))


def hf_vectors(toks):
    out = {}
    for name, tok in toks.items():
        specials = {"llama3_style": ["<|eot_id|>"], "qwen2_style": ["<|im_start|>", "<|im_end|>"], "gpt2": ["<|endoftext|>"]}[name]
        texts = list(TEXTS) + [f"{specials[0]}hello world{specials[-1]}", f"a{specials[0]}b"]
        rows = []
        for t in texts:
            if not t:
                continue
            e = tok.encode(t)
            rows.append({"text": t, "ids": e.ids, "decoded": tok.decode(e.ids, skip_special_tokens=False)})
        out[name] = rows
    return out


def regex_vectors():
    out = []
    for name, pat in (("gpt2", GPT2_PAT), ("llama3", LLAMA3_PAT), ("qwen2", QWEN2_PAT), ("o200k", O200K_PAT), ("dsv3", DSV3_PAT)):
        rx = regex.compile(pat)
        for t in TEXTS:
            b = t.encode("utf-8")
            spans = []
            for m in rx.finditer(t):
                if m.end() > m.start():
                    spans.append([len(t[:m.start()].encode("utf-8")), len(t[:m.end()].encode("utf-8"))])
            out.append({"pattern_name": name, "pattern": pat, "text": t, "spans": spans})
            assert sum(e - s for s, e in spans) <= len(b)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    copy_reference_data()
    with open(os.path.join(OUT, "reference_vectors.json"), "w", encoding="utf-8") as f:
        json.dump(reference_vectors(), f, ensure_ascii=False, indent=1)
    toks = {"gpt2": Tokenizer.from_file(os.path.join(OUT, "gpt2", "tokenizer.json")), "llama3_style": train("llama3"), "qwen2_style": train("qwen2")}
    with open(os.path.join(OUT, "hf_vectors.json"), "w", encoding="utf-8") as f:
        json.dump(hf_vectors(toks), f, ensure_ascii=False, indent=1)
    with open(os.path.join(OUT, "regex_vectors.json"), "w", encoding="utf-8") as f:
        json.dump(regex_vectors(), f, ensure_ascii=False, indent=1)
    rv = reference_vectors()
    print({k: len(v["pairs"]) for k, v in rv.items()}, "reference pairs;", {k: v.get_vocab_size() for k, v in toks.items()}, "vocab sizes")


if __name__ == "__main__":
    sys.exit(main())
