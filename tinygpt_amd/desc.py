"""Model description for the decode path: the Python mirror of ``tgx_model_desc`` (include/tgx.h).

The field set and the parsing rules restate what the reference reads from an HF ``config.json``
(reference: src/huggingface/ModelConfig.cpp:43-125) and how each family factory turns it into
layer hyper-parameters (src/model/ModelLlama.h:21-53, ModelQwen2.h:23-45, ModelMistral.h:23-40,
ModelGPT2.h:226-230).  Both the hub-era flat keys the reference parses (``rope_theta``,
``rope_scaling``, ``torch_dtype``) and the nested ``rope_parameters`` / ``dtype`` keys newer
``transformers`` writes are accepted.
"""
from __future__ import annotations

import ctypes
import json
import os
from dataclasses import dataclass, field, asdict

# tgx_family (include/tgx.h)
FAMILY_GPT2, FAMILY_LLAMA, FAMILY_QWEN2, FAMILY_QWEN3, FAMILY_MISTRAL = 1, 2, 3, 4, 5
FAMILY_BY_NAME = {"gpt2": FAMILY_GPT2, "llama": FAMILY_LLAMA, "qwen2": FAMILY_QWEN2,
                  "qwen3": FAMILY_QWEN3, "mistral": FAMILY_MISTRAL}
# tgx_dtype (include/tgx.h)
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
DTYPE_BY_NAME = {"float32": DTYPE_F32, "fp32": DTYPE_F32, "bfloat16": DTYPE_BF16, "bf16": DTYPE_BF16,
                 "float16": DTYPE_F16, "fp16": DTYPE_F16}


class CDesc(ctypes.Structure):
    """ctypes image of ``struct tgx_model_desc`` — keep in lock-step with include/tgx.h."""
    _fields_ = [
        ("family", ctypes.c_int32),
        ("hidden", ctypes.c_int32),
        ("layers", ctypes.c_int32),
        ("heads", ctypes.c_int32),
        ("kv_heads", ctypes.c_int32),
        ("head_dim", ctypes.c_int32),
        ("inter", ctypes.c_int32),
        ("vocab", ctypes.c_int32),
        ("max_ctx", ctypes.c_int32),
        ("qkv_bias", ctypes.c_int32),
        ("tied", ctypes.c_int32),
        ("compute_dtype", ctypes.c_int32),
        ("norm_eps", ctypes.c_float),
        ("rope_theta", ctypes.c_float),
        ("rope_factor", ctypes.c_float),
        ("rope_low_freq", ctypes.c_float),
        ("rope_high_freq", ctypes.c_float),
        ("rope_orig_ctx", ctypes.c_int32),
        ("n_positions", ctypes.c_int32),
        ("max_batch", ctypes.c_int32),
        ("qk_norm", ctypes.c_int32),
    ]


@dataclass
class ModelDesc:
    family: str = "llama"
    hidden: int = 0
    layers: int = 0
    heads: int = 0
    kv_heads: int = 0
    head_dim: int = 0
    inter: int = 0
    vocab: int = 0
    max_ctx: int = 0            # == GPTModel::contextSize(): KV-cache capacity and RoPE table length
    qkv_bias: bool = False
    tied: bool = False
    compute_dtype: str = "bf16"  # the CLI's --dtype; weights are cast to it after load
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_factor: float = 0.0     # 0 => no llama3 scaling (std::nullopt in Qwen2/Mistral factories)
    rope_low_freq: float = 1.0
    rope_high_freq: float = 1.0
    rope_orig_ctx: int = 0
    n_positions: int = 0         # GPT-2 learned position table rows
    max_batch: int = 1
    qk_norm: bool = False        # Qwen3: per-head RMSNorm on q, k before RoPE (Attention.h:128-167)
    name: str = ""               # label only (bench/config reporting)

    # ---- derived sizes -------------------------------------------------------------------
    @property
    def q_dim(self): return self.heads * self.head_dim

    @property
    def kv_dim(self): return self.kv_heads * self.head_dim

    def to_c(self) -> CDesc:
        return CDesc(FAMILY_BY_NAME[self.family], self.hidden, self.layers, self.heads, self.kv_heads,
                     self.head_dim, self.inter, self.vocab, self.max_ctx, int(self.qkv_bias), int(self.tied),
                     DTYPE_BY_NAME[self.compute_dtype], self.norm_eps, self.rope_theta, self.rope_factor,
                     self.rope_low_freq, self.rope_high_freq, self.rope_orig_ctx, self.n_positions,
                     self.max_batch, int(self.qk_norm))

    def to_dict(self):
        return asdict(self)

    # ---- tensor inventory (HF checkpoint names the reference loads by, SafeTensors.cpp:157-215) ----
    def tensor_shapes(self) -> "dict[str, tuple]":
        H, I, V = self.hidden, self.inter, self.vocab
        out = {}
        if self.family == "gpt2":
            # hub layout without the "transformer." prefix (ModelGPT2.h:226)
            out["wte.weight"] = (V, H)
            out["wpe.weight"] = (self.n_positions, H)
            for i in range(self.layers):
                p = f"h.{i}."
                out[p + "ln_1.weight"] = (H,); out[p + "ln_1.bias"] = (H,)
                out[p + "attn.c_attn.weight"] = (H, 3 * H); out[p + "attn.c_attn.bias"] = (3 * H,)
                out[p + "attn.c_proj.weight"] = (H, H); out[p + "attn.c_proj.bias"] = (H,)
                out[p + "ln_2.weight"] = (H,); out[p + "ln_2.bias"] = (H,)
                out[p + "mlp.c_fc.weight"] = (H, 4 * H); out[p + "mlp.c_fc.bias"] = (4 * H,)
                out[p + "mlp.c_proj.weight"] = (4 * H, H); out[p + "mlp.c_proj.bias"] = (H,)
            out["ln_f.weight"] = (H,); out["ln_f.bias"] = (H,)
            return out
        out["model.embed_tokens.weight"] = (V, H)
        for i in range(self.layers):
            p = f"model.layers.{i}."
            out[p + "input_layernorm.weight"] = (H,)
            out[p + "self_attn.q_proj.weight"] = (self.q_dim, H)
            out[p + "self_attn.k_proj.weight"] = (self.kv_dim, H)
            out[p + "self_attn.v_proj.weight"] = (self.kv_dim, H)
            if self.qkv_bias:
                out[p + "self_attn.q_proj.bias"] = (self.q_dim,)
                out[p + "self_attn.k_proj.bias"] = (self.kv_dim,)
                out[p + "self_attn.v_proj.bias"] = (self.kv_dim,)
            if self.qk_norm:
                out[p + "self_attn.q_norm.weight"] = (self.head_dim,)
                out[p + "self_attn.k_norm.weight"] = (self.head_dim,)
            out[p + "self_attn.o_proj.weight"] = (H, self.q_dim)
            out[p + "post_attention_layernorm.weight"] = (H,)
            out[p + "mlp.gate_proj.weight"] = (I, H)
            out[p + "mlp.up_proj.weight"] = (I, H)
            out[p + "mlp.down_proj.weight"] = (H, I)
        out["model.norm.weight"] = (H,)
        if not self.tied:
            out["lm_head.weight"] = (V, H)
        return out

    def param_count(self) -> int:
        n = 0
        for s in self.tensor_shapes().values():
            k = 1
            for d in s:
                k *= d
            n += k
        return n

    # ---- algorithmic HBM bytes per decoded token at context T ------------
    def bytes_per_token(self, T: int, elem: int = 2) -> int:
        H, I, V, L = self.hidden, self.inter, self.vocab, self.layers
        q, kv = self.q_dim, self.kv_dim
        if self.family == "gpt2":   # c_attn, c_proj, c_fc, mlp.c_proj with biases, two LayerNorms (w + b); ln_f, one wpe row, the wte head
            per_layer = 3 * H * H + 3 * H + H * H + H + I * H + I + H * I + H + 4 * H
            return elem * (L * per_layer + 2 * H + H + V * H) + elem * 2 * L * kv * T
        per_layer = (q + 2 * kv) * H + ((q + 2 * kv) if self.qkv_bias else 0) + H * q + 2 * I * H + H * I + 2 * H
        return elem * (L * per_layer + H + V * H) + elem * 2 * L * kv * T


def desc_from_hf_config(cfg: dict, compute_dtype: str = "bf16", max_batch: int = 1) -> ModelDesc:
    """config.json dict -> ModelDesc, following ModelConfig.cpp:63-122 and the family factories."""
    mt = cfg.get("model_type", "")
    if mt not in FAMILY_BY_NAME:
        raise ValueError(f"Unsupported model_type: {mt!r}")          # ModelConfig.cpp:108-110
    if mt == "gpt2":
        E = int(cfg.get("n_embd", -1))
        return ModelDesc(family="gpt2", hidden=E, layers=int(cfg.get("n_layer", -1)), heads=int(cfg.get("n_head", -1)),
                         kv_heads=int(cfg.get("n_head", -1)), head_dim=E // int(cfg.get("n_head", 1)), inter=4 * E,
                         vocab=int(cfg.get("vocab_size", -1)),
                         max_ctx=int(cfg.get("n_ctx", cfg.get("n_positions", -1))),   # contextSize = n_ctx (ModelGPT2.h:230)
                         qkv_bias=True, tied=True, compute_dtype=compute_dtype,
                         norm_eps=float(cfg.get("layer_norm_epsilon", 1e-5)), rope_theta=0.0,
                         n_positions=int(cfg.get("n_positions", -1)), max_batch=max_batch, name=cfg.get("_name_or_path", ""))
    rp = cfg.get("rope_parameters") or {}
    rs = cfg.get("rope_scaling") or {}
    if not rs and rp.get("rope_type", "default") not in ("default", None):
        rs = rp
    default_theta = 1.0 if mt == "llama" else 10000.0                  # ModelConfig.cpp:88,91,99
    theta = float(cfg.get("rope_theta", rp.get("rope_theta", default_theta)))
    heads = int(cfg["num_attention_heads"])
    hidden = int(cfg["hidden_size"])
    head_dim = hidden // heads                                         # ModelLlama.h:37 ignores "head_dim"
    if mt == "qwen3":
        head_dim = int(cfg.get("head_dim", head_dim))                  # ModelQwen3.h:25
    d = ModelDesc(family=mt, hidden=hidden, layers=int(cfg["num_hidden_layers"]), heads=heads,
                  kv_heads=int(cfg.get("num_key_value_heads", heads)), head_dim=head_dim,
                  inter=int(cfg["intermediate_size"]), vocab=int(cfg["vocab_size"]),
                  max_ctx=int(cfg.get("max_position_embeddings", -1)),
                  qkv_bias=(mt == "qwen2"),                            # ModelQwen2.h:26-31
                  tied=bool(cfg.get("tie_word_embeddings", False)),
                  compute_dtype=compute_dtype, norm_eps=float(cfg.get("rms_norm_eps", 1e-5)),
                  rope_theta=theta, max_batch=max_batch, qk_norm=(mt == "qwen3"), name=cfg.get("_name_or_path", ""))
    if mt == "llama" and rs:
        d.rope_factor = float(rs.get("factor", 1.0))
        d.rope_high_freq = float(rs.get("high_freq_factor", 1.0))
        d.rope_low_freq = float(rs.get("low_freq_factor", 1.0))
        d.rope_orig_ctx = int(rs.get("original_max_position_embeddings", -1))
        if d.rope_orig_ctx > 0:
            d.max_ctx = d.rope_orig_ctx                                # ModelLlama.h:26-31
    return d


def load_desc(model_dir: str, compute_dtype: str = "bf16", max_batch: int = 1) -> ModelDesc:
    with open(os.path.join(model_dir, "config.json")) as f:
        return desc_from_hf_config(json.load(f), compute_dtype, max_batch)


# ---- the public HF hyper-parameters of BASELINE.json's configs (SURVEY.md §8 header) --------
def _llama(name, H, L, nh, nkv, I, V, tied, theta=500000.0, scaled=True, max_pos=131072):
    cfg = {"model_type": "llama", "hidden_size": H, "num_hidden_layers": L, "num_attention_heads": nh,
           "num_key_value_heads": nkv, "intermediate_size": I, "vocab_size": V, "tie_word_embeddings": tied,
           "rms_norm_eps": 1e-5, "rope_theta": theta, "max_position_embeddings": max_pos, "torch_dtype": "bfloat16",
           "_name_or_path": name}
    if scaled:
        cfg["rope_scaling"] = {"factor": 32.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                               "original_max_position_embeddings": 8192, "rope_type": "llama3"}
    return cfg


KNOWN_CONFIGS = {
    "llama-3.2-1b": _llama("Llama-3.2-1B", 2048, 16, 32, 8, 8192, 128256, True),
    "llama-3.2-3b": _llama("Llama-3.2-3B", 3072, 28, 24, 8, 8192, 128256, True),
    "qwen2.5-0.5b": {"model_type": "qwen2", "hidden_size": 896, "num_hidden_layers": 24, "num_attention_heads": 14,
                     "num_key_value_heads": 2, "intermediate_size": 4864, "vocab_size": 151936,
                     "tie_word_embeddings": True, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                     "max_position_embeddings": 32768, "torch_dtype": "bfloat16", "_name_or_path": "Qwen2.5-0.5B"},
    "mistral-7b-v0.3": {"model_type": "mistral", "hidden_size": 4096, "num_hidden_layers": 32,
                        "num_attention_heads": 32, "num_key_value_heads": 8, "intermediate_size": 14336,
                        "vocab_size": 32768, "tie_word_embeddings": False, "rms_norm_eps": 1e-5,
                        "rope_theta": 1000000.0, "max_position_embeddings": 32768, "torch_dtype": "bfloat16",
                        "_name_or_path": "Mistral-7B-v0.3"},
    # the other checkpoints the reference's README names (README.md:50-56), public config.json values
    "qwen2.5-3b": {"model_type": "qwen2", "hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16,
                   "num_key_value_heads": 2, "intermediate_size": 11008, "vocab_size": 151936,
                   "tie_word_embeddings": True, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                   "max_position_embeddings": 32768, "torch_dtype": "bfloat16", "_name_or_path": "Qwen2.5-3B"},
    "qwen3-1.7b": {"model_type": "qwen3", "hidden_size": 2048, "num_hidden_layers": 28, "num_attention_heads": 16,
                   "num_key_value_heads": 8, "head_dim": 128, "intermediate_size": 6144, "vocab_size": 151936,
                   "tie_word_embeddings": True, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                   "max_position_embeddings": 40960, "torch_dtype": "bfloat16", "attention_bias": False,
                   "_name_or_path": "Qwen3-1.7B"},
    # 70B-class geometry (Llama-3.1-70B): 141 GB of bf16 parameters — one MI355X holds it in its 288 GB of HBM3E
    "llama-3.1-70b": {"model_type": "llama", "hidden_size": 8192, "num_hidden_layers": 80, "num_attention_heads": 64,
                      "num_key_value_heads": 8, "intermediate_size": 28672, "vocab_size": 128256,
                      "tie_word_embeddings": False, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                      "max_position_embeddings": 131072, "torch_dtype": "bfloat16", "_name_or_path": "Llama-3.1-70B",
                      "rope_scaling": {"factor": 8.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                                       "original_max_position_embeddings": 8192, "rope_type": "llama3"}},
    # Qwen3-0.6B (public config.json): per-head q/k RMSNorm, explicit head_dim 128 with q_dim 2048 != hidden 1024
    "qwen3-0.6b": {"model_type": "qwen3", "hidden_size": 1024, "num_hidden_layers": 28, "num_attention_heads": 16,
                   "num_key_value_heads": 8, "head_dim": 128, "intermediate_size": 3072, "vocab_size": 151936,
                   "tie_word_embeddings": True, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                   "max_position_embeddings": 40960, "torch_dtype": "bfloat16", "attention_bias": False,
                   "_name_or_path": "Qwen3-0.6B"},
    "gpt2": {"model_type": "gpt2", "n_embd": 768, "n_layer": 12, "n_head": 12, "n_ctx": 1024, "n_positions": 1024,
             "vocab_size": 50257, "layer_norm_epsilon": 1e-5, "activation_function": "gelu_new",
             "torch_dtype": "float32", "_name_or_path": "gpt2"},
}


def known_desc(key: str, compute_dtype: str | None = None, max_batch: int = 1) -> ModelDesc:
    cfg = KNOWN_CONFIGS[key.lower()]
    if compute_dtype is None:
        compute_dtype = "fp32" if cfg["model_type"] == "gpt2" else "bf16"
    return desc_from_hf_config(cfg, compute_dtype, max_batch)
