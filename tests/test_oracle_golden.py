"""The CPU oracle against the golden vectors from HF transformers (tools/gen_fixtures.py).

fp32 mode pins structure (RoPE pairing and llama3 scaling, GQA mapping, merge order, norms, GPT-2's
Conv1D/LayerNorm/gelu_new/left-pad-no-mask) at 1e-4 relative.  bf16 mode (bf16 parameters + bf16 KV cache,
fp32 activations — DESIGN.md §3) is held to HF-fp32 within 2e-2 (the only difference is the KV rounding;
measured 1.6e-3..1.4e-2 on these vectors) and to HF-bf16 within 8e-2 (HF-bf16 itself sits 0.7-4.4e-2 from
HF-fp32), with greedy ids equal to HF's in both dtypes.  fp16 mode (half parameters + half KV cache) is held to
HF-fp32 within 2e-3 (measured <= 8e-4) and to HF-fp16 (golden_fp16.npz) within 8e-3 (measured <= 3.3e-3).
"""
import os
import numpy as np
import pytest

from conftest import FAMILIES, load_golden, rel_err
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY

TOL = {"fp32": 1e-4, "bf16": 2e-2}      # vs HF fp32 logits
TOL_BF16_VS_HF_BF16 = 8e-2


def make_oracle(fam, mode, oracle_lib):
    from oracle.oracle_ffi import OracleModel
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, mode, max_batch=g["prompt"].shape[0])
    m = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    return m, g


@pytest.mark.parametrize("fam", FAMILIES)
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_free_running_greedy_matches_hf(fam, mode, oracle_lib):
    """prefill + 15 decode steps == generateSync; ids identical, logits within tolerance."""
    m, g = make_oracle(fam, mode, oracle_lib)
    L, ids = g["logits_fp32"], g[f"ids_{mode}"]
    m.forward(g["prompt"])
    assert m.past_length == g["prompt"].shape[1]
    assert rel_err(m.logits(rounded=False), L[:, 0]) < TOL[mode]
    if mode == "bf16":
        assert rel_err(m.logits(rounded=False), g["logits_bf16"][:, 0]) < TOL_BF16_VS_HF_BF16
    first = m.sample(GREEDY)
    np.testing.assert_array_equal(first, ids[:, 0])
    rest = m.decode(L.shape[1] - 1, GREEDY)                  # [n-1, B]
    np.testing.assert_array_equal(rest.T, ids[:, 1:])
    assert rel_err(m.logits(rounded=False), L[:, -1]) < TOL[mode]
    assert m.past_length == g["prompt"].shape[1] + L.shape[1] - 1


@pytest.mark.parametrize("fam", FAMILIES)
def test_fp16_mode_matches_hf(fam, oracle_lib):
    if fam == "gpt2_tiny":
        pytest.skip("the GPT-2 fixture is the reference's fp32 CPU case")
    from conftest import GOLDEN
    m, g = make_oracle(fam, "fp16", oracle_lib)
    g16 = np.load(os.path.join(GOLDEN, fam, "golden_fp16.npz"))
    m.forward(g["prompt"])
    assert rel_err(m.logits(rounded=False), g["logits_fp32"][:, 0]) < 2e-3
    assert rel_err(m.logits(rounded=False), g16["logits_fp16"][:, 0]) < 8e-3
    np.testing.assert_array_equal(m.sample(GREEDY), g16["ids_fp16"][:, 0])
    rest = m.decode(g16["ids_fp16"].shape[1] - 1, GREEDY)
    np.testing.assert_array_equal(rest.T, g16["ids_fp16"][:, 1:])
    assert rel_err(m.logits(rounded=False), g["logits_fp32"][:, -1]) < 2e-3
    assert rel_err(m.logits(rounded=False), g16["logits_fp16"][:, -1]) < 8e-3


@pytest.mark.parametrize("fam", FAMILIES)
def test_teacher_forced_logits_every_step(fam, oracle_lib):
    m, g = make_oracle(fam, "fp32", oracle_lib)
    L, ids = g["logits_fp32"], g["ids_fp32"]
    m.forward(g["prompt"])
    for i in range(1, L.shape[1]):
        m.forward(ids[:, i - 1:i])
        assert rel_err(m.logits(rounded=False), L[:, i]) < 1e-4, f"step {i}"


@pytest.mark.parametrize("fam", [f for f in FAMILIES if not f.startswith("gpt2")])
def test_rope_tables(fam, oracle_lib):
    m, g = make_oracle(fam, "fp32", oracle_lib)
    cos, sin = m.rope_tables(g["rope_cos"].shape[0])
    assert np.abs(cos - g["rope_cos"]).max() < 1e-6
    assert np.abs(sin - g["rope_sin"]).max() < 1e-6


def test_kv_cache_grows_by_append(oracle_lib):
    """KVCacheManager semantics (CacheManager.h:24-51): rows already cached never change; reset empties."""
    m, g = make_oracle("llama_tiny", "bf16", oracle_lib)
    m.forward(g["prompt"])
    k0, v0 = m.read_kv(0, 1)
    assert k0.shape[0] == g["prompt"].shape[1]
    m.sample(GREEDY)
    m.decode(3, GREEDY)
    k1, v1 = m.read_kv(0, 1)
    assert k1.shape[0] == k0.shape[0] + 3
    np.testing.assert_array_equal(k1[:k0.shape[0]], k0)
    np.testing.assert_array_equal(v1[:v0.shape[0]], v0)
    m.reset_cache()
    assert m.past_length == 0


def test_context_limit_and_bad_calls(oracle_lib):
    from tinygpt_amd.ffi import TgxError
    m, g = make_oracle("llama_tiny", "fp32", oracle_lib)
    assert m.context_size == 64                                # original_max_position_embeddings (ModelLlama.h:26-31)
    with pytest.raises(TgxError):
        m.forward(np.zeros((1, 65), np.int64))
    m.forward(g["prompt"])
    with pytest.raises(TgxError):
        m.forward(g["prompt"])                                 # seq>1 with pastLength>0
    with pytest.raises(TgxError):
        m.forward(np.array([[10_000]]))                        # id out of range


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("reorder", [False, True])
@pytest.mark.parametrize("fam", ["llama_tiny", "qwen2_tiny", "gpt2_hd64"])
def test_four_row_pass_is_bit_identical(fam, mode, reorder, oracle_lib):
    """linear() takes four weight rows per pass over k (shared activation loads, four overlapping add chains); each row's sum is the one dot_row()
    forms, in its order: logits and cache rows bit-identical to the row-by-row form, forwards and under tgxo_set_reorder, prompt and steps."""
    outs = []
    for one_row in (1, 0):
        oracle_lib.set_one_row_dots(one_row)
        try:
            m, g = make_oracle(fam, mode, oracle_lib)
            m.set_reorder(reorder)
            m.forward(g["prompt"]); m.sample(GREEDY); m.decode(3, GREEDY)
            outs.append((m.logits(rounded=False).copy(), [m.read_kv(0, layer) for layer in range(m.desc.layers)]))
        finally:
            oracle_lib.set_one_row_dots(0)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for (k0, v0), (k1, v1) in zip(outs[0][1], outs[1][1]):
        np.testing.assert_array_equal(k0, k1); np.testing.assert_array_equal(v0, v1)


@pytest.mark.parametrize("fam", [f for f in FAMILIES if f != "gpt2_tiny"])
def test_torch_rounding_contract_vs_hf_bf16(fam, oracle_lib):
    """The reference's --dtype bf16 builds every module in bf16 (src/model/ModelLlama.h:62, src/huggingface/ModelLoader.cpp:84): each op's OUTPUT is a bf16
    tensor.  The oracle restates that contract op by op (tgxo_set_torch_rounding); HF-bf16 (`logits_bf16`, `ids_bf16`) is the only executable
    instance of it in reach (TinyTorch is absent), so this is where "where a bf16 module rounds" is pinned: teacher-forced with HF-bf16's ids over the
    prompt + 15 steps, the per-op-rounded oracle stays inside bf16's own schedule-vs-schedule floor of HF-bf16 (two implementations that round at the
    same points but sum in another order agree only to 3-6e-3 per flip; measured 0.8-3.6e-2 here, the default fp32-activation contract 0.6-5.3e-2),
    every argmax equal to HF-bf16's.  It is FARTHER from HF-fp32 than the default contract (1.1-4.6e-2 against 0.2-1.4e-2): rounding between ops is what
    that contract costs.  The table over all three contracts and both sides: tools/contracts_table.py -> profiles/r05_contracts.txt."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from contracts_table import contracts
    rows = {mode: (e16, e32, same, n) for side, mode, e16, e32, same, n in contracts(fam)}
    t16, t32, same, n = rows["torch-rounding"]
    d16, d32, dsame, _ = rows["fp32-act"]
    a16, a32, asame, _ = rows["act.round16"]
    print(f"{fam}: vs HF-bf16 / HF-fp32  fp32-act {d16:.2e} / {d32:.2e}   act.round16 {a16:.2e} / {a32:.2e}   torch-rounding {t16:.2e} / {t32:.2e}")
    assert same == n and dsame == n and asame == n            # greedy ids identical to HF-bf16 under all three contracts
    assert t16 < TOL_BF16_VS_HF_BF16 and d16 < TOL_BF16_VS_HF_BF16 and a16 < TOL_BF16_VS_HF_BF16
    assert d32 < TOL["bf16"]                                   # the default contract is the one held to HF-fp32
    assert d32 < a32 < 2 * TOL["bf16"] and d32 < t32 < 3 * TOL["bf16"]      # each rounding added moves AWAY from the fp32 model: fp32-act < act.round16 (< torch-rounding on most)
