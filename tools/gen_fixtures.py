#!/usr/bin/env python3
"""Generate tests/golden/* — golden vectors for the decode path from HF transformers (CPU).

Runs ONLY in the build container (torch + transformers importable); nothing here travels to the GPU
box except the small .npz/.json files it writes.  The reference's own arithmetic (TinyTorch) is
absent, so HF `transformers` — whose checkpoints the reference loads by tensor name — is the upstream
truth these vectors pin.

For each tiny family config:
  * weights: tinygpt_amd.synth (deterministic integer hash, bf16-exact) loaded into the HF model;
  * HF eager attention, fp32 and bf16, prefill of a 9-token prompt + 16 greedy steps with KV cache;
  * stored: prompt, per-step greedy ids and last-position logits, last-position hidden states of
    the prefill, RoPE cos/sin tables, and the top-2 logit gap of every greedy step (the seed is
    bumped until no bf16 near-tie occurs on the greedy path, so "greedy ids identical" is a
    well-posed test);
  * GPT-2: batch 4, left-padded with id 0, no attention mask (GPTEngine.cpp:95,108-138).
Also writes sampler vectors: Sampler.cpp:34-77 restated 1:1 with torch ops (sort/softmax/cumsum/
scatter), since HF's own top-p warper keeps a different set (it includes the crossing token).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import synth  # noqa: E402
from tinygpt_amd.desc import desc_from_hf_config  # noqa: E402

import transformers  # noqa: E402
from transformers import (GPT2Config, GPT2LMHeadModel, LlamaConfig, LlamaForCausalLM, MistralConfig,  # noqa: E402
                          MistralForCausalLM, Qwen2Config, Qwen2ForCausalLM, Qwen3Config, Qwen3ForCausalLM)

OUT = os.path.join(ROOT, "tests", "golden")
N_PROMPT, N_STEPS, STD = 9, 16, 0.08

# hub-era flat config.json dicts (the schema the reference parses, ModelConfig.cpp:63-122)
FAMILIES = {
    "llama_tiny": {
        "model_type": "llama", "hidden_size": 256, "num_hidden_layers": 2, "num_attention_heads": 4,
        "num_key_value_heads": 2, "intermediate_size": 512, "vocab_size": 256, "tie_word_embeddings": True,
        "rms_norm_eps": 1e-5, "rope_theta": 500000.0, "max_position_embeddings": 256, "torch_dtype": "bfloat16",
        "hidden_act": "silu", "attention_bias": False, "bos_token_id": 1, "eos_token_id": 2,
        "rope_scaling": {"factor": 32.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                         "original_max_position_embeddings": 64, "rope_type": "llama3"}},
    "qwen2_tiny": {
        "model_type": "qwen2", "hidden_size": 192, "num_hidden_layers": 2, "num_attention_heads": 3,
        "num_key_value_heads": 1, "intermediate_size": 320, "vocab_size": 320, "tie_word_embeddings": True,
        "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "max_position_embeddings": 128, "torch_dtype": "bfloat16",
        "hidden_act": "silu", "bos_token_id": 1, "eos_token_id": 2, "use_sliding_window": False},
    "mistral_tiny": {
        "model_type": "mistral", "hidden_size": 512, "num_hidden_layers": 2, "num_attention_heads": 4,
        "num_key_value_heads": 2, "intermediate_size": 640, "vocab_size": 256, "tie_word_embeddings": False,
        "rms_norm_eps": 1e-5, "rope_theta": 1000000.0, "max_position_embeddings": 128, "torch_dtype": "bfloat16",
        "hidden_act": "silu", "bos_token_id": 1, "eos_token_id": 2, "sliding_window": None},
    "qwen3_tiny": {   # per-head q/k RMSNorm, explicit head_dim with q_dim (256) != hidden_size (192)
        "model_type": "qwen3", "hidden_size": 192, "num_hidden_layers": 2, "num_attention_heads": 4,
        "num_key_value_heads": 2, "head_dim": 64, "intermediate_size": 320, "vocab_size": 320,
        "tie_word_embeddings": True, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "max_position_embeddings": 128,
        "torch_dtype": "bfloat16", "hidden_act": "silu", "bos_token_id": 1, "eos_token_id": 2,
        "use_sliding_window": False, "attention_bias": False},
    "gpt2_tiny": {
        "model_type": "gpt2", "n_embd": 64, "n_layer": 2, "n_head": 2, "n_ctx": 64, "n_positions": 64,
        "vocab_size": 256, "layer_norm_epsilon": 1e-5, "activation_function": "gelu_new", "torch_dtype": "float32",
        "bos_token_id": 1, "eos_token_id": 2},
    "gpt2_hd64": {   # head_dim 64 (every released GPT-2 size has it): the geometry the MI355X kernels cover; 24 weight slices per row
        "model_type": "gpt2", "n_embd": 192, "n_layer": 2, "n_head": 3, "n_ctx": 64, "n_positions": 64,
        "vocab_size": 320, "layer_norm_epsilon": 1e-5, "activation_function": "gelu_new", "torch_dtype": "float32",
        "bos_token_id": 1, "eos_token_id": 2},
}
HF_CLASSES = {"llama": (LlamaConfig, LlamaForCausalLM), "qwen2": (Qwen2Config, Qwen2ForCausalLM),
              "mistral": (MistralConfig, MistralForCausalLM), "gpt2": (GPT2Config, GPT2LMHeadModel),
              "qwen3": (Qwen3Config, Qwen3ForCausalLM)}


def build_hf(cfg: dict, seed: int, dtype):
    desc = desc_from_hf_config(cfg, "fp32")
    ccls, mcls = HF_CLASSES[cfg["model_type"]]
    kw = {k: v for k, v in cfg.items() if k not in ("model_type", "torch_dtype")}
    hcfg = ccls(**kw, attn_implementation="eager")
    model = mcls(hcfg).eval()
    sd = {}
    for name, bits in synth.synth_checkpoint(desc, seed, STD):
        t = torch.from_numpy(synth.bf16_bits_to_f32(bits).copy())
        if cfg["model_type"] == "gpt2":
            name = "transformer." + name        # save_pretrained layout; the reference expects it without (ModelGPT2.h:226)
        sd[name] = t
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if not (m.endswith("lm_head.weight") or m.endswith(".attn.bias") or m.endswith("masked_bias"))]
    assert not missing and not unexpected, (missing, unexpected)
    if cfg["model_type"] != "gpt2" and hasattr(model.model, "rotary_emb"):
        inv = model.model.rotary_emb.inv_freq.clone()
    model = model.to(dtype)
    if cfg["model_type"] != "gpt2" and hasattr(model.model, "rotary_emb"):
        model.model.rotary_emb.inv_freq = inv            # keep inv_freq fp32 (as from_pretrained does)
        if hasattr(model.model.rotary_emb, "original_inv_freq"):
            model.model.rotary_emb.original_inv_freq = inv
    return desc, model


@torch.no_grad()
def run_greedy(model, ids: torch.Tensor, n_steps: int):
    """prefill + n_steps-1 cached decode steps == generateSync (GPTEngine.cpp:154-174)."""
    out = model(ids, use_cache=True, output_hidden_states=True)
    pkv = out.past_key_values
    hidden = torch.stack([h[:, -1, :].float() for h in out.hidden_states], 1)     # [B, L+1, H]
    logits = [out.logits[:, -1, :].float()]
    toks = [logits[-1].argmax(-1)]
    for _ in range(n_steps - 1):
        o = model(toks[-1][:, None], past_key_values=pkv, use_cache=True)
        pkv = o.past_key_values
        logits.append(o.logits[:, -1, :].float())
        toks.append(logits[-1].argmax(-1))
    return torch.stack(logits, 1).numpy(), torch.stack(toks, 1).numpy(), hidden.numpy()   # [B,n,V], [B,n], [B,L+1,H]


def top2_gap_ulps(logits: np.ndarray) -> float:
    """smallest (top1-top2)/bf16-ulp(top1) over all steps/rows"""
    s = np.sort(logits, -1)
    top1, top2 = s[..., -1], s[..., -2]
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(top1), 1e-30))) - 7)
    return float(((top1 - top2) / ulp).min())


def rope_tables(model, hd, n_pos=32):
    rot = model.model.rotary_emb
    x = torch.zeros(1, 1, hd, dtype=torch.float32)
    cos, sin = rot(x, torch.arange(n_pos)[None, :])
    return cos[0, :, : hd // 2].numpy(), sin[0, :, : hd // 2].numpy()


def gen_family(name: str, cfg: dict):
    gpt2 = cfg["model_type"] == "gpt2"
    seed = 1234
    while True:
        desc, m32 = build_hf(cfg, seed, torch.float32)
        _, m16 = build_hf(cfg, seed, torch.bfloat16)
        if gpt2:
            # the CLI's 4 prompts have lengths 5/7/5/5: left-pad with id 0 to 7, no mask
            lens = [5, 7, 5, 5]
            S = max(lens)
            rows = []
            for r, ln in enumerate(lens):
                p = synth.synth_prompt(cfg["vocab_size"] - 1, ln, seed + r) + 1
                rows.append(np.concatenate([np.zeros(S - ln, np.int64), p]))
            prompt = np.stack(rows)
        else:
            prompt = synth.synth_prompt(cfg["vocab_size"], N_PROMPT, seed)[None, :]
        ids = torch.from_numpy(prompt)
        l32, t32, h32 = run_greedy(m32, ids, N_STEPS)
        l16, t16, h16 = run_greedy(m16, ids, N_STEPS)
        gap32, gap16 = top2_gap_ulps(l32), top2_gap_ulps(l16)
        if gap16 >= 4.0 and gap32 >= 4.0:
            break
        print(f"  {name}: seed {seed} has a near-tie on the greedy path (gap {gap16:.2f}/{gap32:.2f} ulp) -> next seed")
        seed += 1
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    with open(os.path.join(d, "generation_config.json"), "w") as f:
        json.dump({"bos_token_id": cfg.get("bos_token_id", 1), "eos_token_id": cfg.get("eos_token_id", 2)}, f)
    arrs = dict(prompt=prompt, seed=np.int64(seed), std=np.float32(STD),
                logits_fp32=l32, ids_fp32=t32, hidden_fp32=h32,
                logits_bf16=l16, ids_bf16=t16, hidden_bf16=h16,
                gap_ulps=np.float32([gap32, gap16]))
    if not gpt2:
        cos, sin = rope_tables(m32, desc.head_dim)
        arrs.update(rope_cos=cos, rope_sin=sin)
    np.savez_compressed(os.path.join(d, "golden.npz"), **arrs)
    print(f"{name}: seed {seed}, params {desc.param_count()}, greedy fp32 {t32[0][:8]}..., bf16 {t16[0][:8]}..., "
          f"min top-2 gap {gap32:.1f}/{gap16:.1f} bf16-ulps, fp32-vs-bf16 same ids: {bool((t32 == t16).all())}")


def gen_family_fp16(name: str, cfg: dict):
    """fp16 companion of an existing fixture (same seed, same prompt): tests/golden/<name>/golden_fp16.npz.
    Pins the fp16 row of the --dtype matrix (fp16 parameters + fp16 KV cache) without touching golden.npz."""
    g = np.load(os.path.join(OUT, name, "golden.npz"))
    seed = int(g["seed"])
    _, m16 = build_hf(cfg, seed, torch.float16)
    l16, t16, h16 = run_greedy(m16, torch.from_numpy(g["prompt"]), N_STEPS)
    s = np.sort(l16, -1)
    gap = float(((s[..., -1] - s[..., -2]) / (2.0 ** (np.floor(np.log2(np.maximum(np.abs(s[..., -1]), 1e-30))) - 10))).min())
    np.savez_compressed(os.path.join(OUT, name, "golden_fp16.npz"), logits_fp16=l16, ids_fp16=t16, hidden_fp16=h16,
                        gap_ulps=np.float32(gap))
    print(f"{name}: fp16 greedy {t16[0][:8]}..., min top-2 gap {gap:.1f} fp16-ulps, same ids as fp32: "
          f"{bool((t16 == g['ids_fp32']).all())}, max |fp16-fp32| logits {np.abs(l16 - g['logits_fp32']).max():.3e}")


def gen_sampler():
    """Sampler.cpp:34-77 restated op-for-op with torch (fp32), for fixed logits vectors."""
    g = torch.Generator().manual_seed(7)
    V = 512
    cases = []
    base = torch.randn(V, generator=g) * 2.5
    tied = base.clone()
    tied[10] = tied[200] = tied.max() + 0.5          # duplicated maximum: argmax must pick index 10
    flat = torch.zeros(V)
    cfgs = [(0.8, 0, 0.9, 0.0), (0.7, 50, 1.0, 0.0), (1.0, 0, 1.0, 0.05), (0.8, 50, 0.9, 0.05), (0.0, 0, 0.5, 0.0),
            (1.5, 5, 0.3, 0.0), (1.0, 1000, 0.999, 0.0)]
    out = {}
    for li, logits in enumerate([base, tied, flat]):
        for ci, (T, K, P, M) in enumerate(cfgs):
            l = logits.clone()[None, :]
            if T > 0:
                l = l / T
            if K > 0:
                k = min(K, V)
                vals, idx = torch.topk(l, k, -1)
                l = torch.full_like(l, float("-inf")).scatter_(-1, idx, vals)
            if P < 1:
                sl, si = torch.sort(l, dim=-1, descending=True, stable=True)
                probs = torch.softmax(sl, -1)
                cum = torch.cumsum(probs, -1)
                mask = cum <= P
                mask[:, 0] = True
                sl = sl.masked_fill(~mask, float("-inf"))
                l = torch.full_like(l, float("-inf")).scatter_(-1, si, sl)
            if M > 0:
                pr = torch.softmax(l, -1)
                mx = pr.max(-1, keepdim=True).values
                l = l.masked_fill(pr < mx * M, float("-inf"))
            probs = torch.softmax(l, -1)
            out[f"case{li}_{ci}_probs"] = probs[0].numpy()
            out[f"case{li}_{ci}_cfg"] = np.float32([T, K, P, M])
        out[f"logits{li}"] = logits.numpy()
        out[f"argmax{li}"] = np.int64(int(torch.argmax(logits)))
    out["n_logits"], out["n_cfgs"] = np.int64(3), np.int64(len(cfgs))
    os.makedirs(os.path.join(OUT, "sampler"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "sampler", "golden.npz"), **out)
    print("sampler: wrote", len(cfgs) * 3, "cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    print("torch", torch.__version__, "transformers", transformers.__version__)
    only = sys.argv[1:]
    if only and only[0] == "--fp16":          # add golden_fp16.npz next to the existing fixtures (which stay untouched)
        for name, cfg in FAMILIES.items():
            if name != "gpt2_tiny" and (len(only) == 1 or name in only[1:]):      # gpt2_tiny is the reference's fp32 CPU case
                gen_family_fp16(name, cfg)
        sys.exit(0)
    for name, cfg in FAMILIES.items():
        if not only or name in only:
            gen_family(name, cfg)
    if not only or "sampler" in only:
        gen_sampler()
