#!/usr/bin/env python3
"""Pin the oracle at the REAL Llama-3.2-1B geometry (16 layers, 32/8 heads of 64, hidden 2048, intermediate 8192, vocabulary
128 256, tied head, llama3 RoPE scaling 32 / 8192): HF transformers on CPU (fp32) with the deterministic synthetic checkpoint,
a 12-token prompt and 3 teacher-forced steps.  Stored (tests/golden/llama_3_2_1b_full/golden.npz, ~30 KB): the prompt, the
forced ids and, per position, the top-64 logits (values + indices) and 64 fixed probe indices — enough to pin magnitude,
ordering and argmax without shipping 128 256 floats per position.  Runs only in the build container (needs ~12 GB RAM)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import synth  # noqa: E402
from tinygpt_amd.desc import KNOWN_CONFIGS, desc_from_hf_config  # noqa: E402
from transformers import (GPT2Config, GPT2LMHeadModel, LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM,  # noqa: E402
                          Qwen3Config, Qwen3ForCausalLM)

SEED, STD = 1234, 0.02


MODELS = {"llama-3.2-1b": (LlamaConfig, LlamaForCausalLM, "llama_3_2_1b_full"), "qwen2.5-0.5b": (Qwen2Config, Qwen2ForCausalLM, "qwen2_5_0_5b_full"),
          "qwen3-0.6b": (Qwen3Config, Qwen3ForCausalLM, "qwen3_0_6b_full"), "gpt2": (GPT2Config, GPT2LMHeadModel, "gpt2_124m_full")}


def main(key="llama-3.2-1b"):
    ccls, mcls, out_name = MODELS[key]
    cfg = dict(KNOWN_CONFIGS[key])
    desc = desc_from_hf_config(cfg, "fp32")
    kw = {k: v for k, v in cfg.items() if k not in ("model_type", "torch_dtype", "_name_or_path", "n_ctx")}
    gpt2 = cfg["model_type"] == "gpt2"
    torch.set_num_threads(8)
    with torch.device("meta"):
        model = mcls(ccls(**kw, attn_implementation="eager"))
    model = model.to_empty(device="cpu").eval()
    sd = {}
    for name, bits in synth.synth_checkpoint(desc, SEED, STD):
        sd[("transformer." + name) if gpt2 else name] = torch.from_numpy(synth.bf16_bits_to_f32(bits).copy())
    missing, unexpected = model.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and all(m.endswith("lm_head.weight") or m.endswith(".attn.bias") or m.endswith(".attn.masked_bias") for m in missing), (missing, unexpected)
    model.tie_weights()
    if gpt2:     # the causal-mask buffers are non-persistent too
        for blk in model.transformer.h:
            if hasattr(blk.attn, "bias") and isinstance(blk.attn.bias, torch.Tensor):
                n = model.config.n_positions
                blk.attn.bias = torch.tril(torch.ones((n, n), dtype=torch.bool)).view(1, 1, n, n)
    else:
        # rotary inv_freq is a non-persistent buffer: rebuild it (to_empty left it uninitialised)
        model.model.rotary_emb = type(model.model.rotary_emb)(config=model.config)
    prompt = synth.synth_prompt(desc.vocab, 12, SEED)[None, :]
    rng = np.random.default_rng(7)
    probe = np.sort(rng.choice(desc.vocab, 64, replace=False)).astype(np.int64)
    forced, top_v, top_i, probe_v = [], [], [], []
    with torch.no_grad():
        out = model(torch.from_numpy(prompt), use_cache=True)
        pkv = out.past_key_values
        for step in range(4):
            l = out.logits[0, -1].float().numpy()
            order = np.argsort(-l, kind="stable")[:64]
            top_i.append(order.astype(np.int64)); top_v.append(l[order]); probe_v.append(l[probe])
            if step == 3:
                break
            tok = int(order[0])
            forced.append(tok)
            out = model(torch.tensor([[tok]]), past_key_values=pkv, use_cache=True)
            pkv = out.past_key_values
    d = os.path.join(ROOT, "tests", "golden", out_name)
    os.makedirs(d, exist_ok=True)
    np.savez_compressed(os.path.join(d, "golden.npz"), prompt=prompt, forced=np.int64(forced), top_v=np.float32(top_v), top_i=np.int64(top_i),
                        probe=probe, probe_v=np.float32(probe_v), seed=np.int64(SEED), std=np.float32(STD))
    print("forced ids", forced, "top-2 gaps", [float(v[0] - v[1]) for v in top_v], "max |logit|", float(np.abs(top_v).max()))


if __name__ == "__main__":
    for k in (sys.argv[1:] or list(MODELS)):
        main(k)
