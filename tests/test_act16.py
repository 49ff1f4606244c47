"""Option act.round16 (round 4): the reference's 16-bit-module contract for the input of a Linear.

TinyGPT constructs its modules in config.torch_dtype (src/model/ModelLlama.h:62): an nn::Linear of a bf16 model multiplies a bf16 tensor.  The default
contract here keeps activations in fp32 and pays for it on the matrix cores with two or three 16-bit terms per activation.  With
tgx_set_option("act.round16", 1) the input of EVERY Linear (qkv, o_proj, gate_up / c_fc, down / c_proj, lm_head) is rounded to the storage dtype first,
round-to-nearest-even, and nothing else changes; the oracle restates it with tgxo_set_act16 (one rounding loop in linear()).

What can be held how tightly: both sides round the same quantities, but a quantity that two fp32 schedules compute one fp32 ulp apart lands on different
16-bit neighbours when it straddles a rounding boundary (~2^-16 of the inputs per Linear) — the same flip mechanism as the bf16 KV cache
(tests/test_oracle_reorder.py), now at every Linear input.  Measured on the two-layer fixtures, oracle against reordered oracle in this mode: 1.8e-7 where no input
flips (qwen2_tiny, qwen3_tiny), 2.1e-3 / 4.9e-3 where one does (llama_tiny, mistral_tiny) — while the two CONTRACTS differ by 4e-3 .. 1.4e-2, so a single
comparison at a floor-sized tolerance could not tell "rounded everywhere" from "one Linear forgot to round".  Hence two kinds of test: (a) where no flip
has happened yet: the second layer's cache row of the FIRST token (the image of that token's whole first layer, and of nothing else) must be
bit-identical in at least 4 of 16 prompts on every prefill and decode path — a Linear that skipped its rounding would move them in every prompt; (b) the fixtures / real geometries / full depth against the floor-sized bound, the flip floor of the mode printed
next to it (oracle vs reordered oracle).  Forms that must not change results inside the mode are held to BIT-IDENTITY (one-term kernels vs two-term kernels fed zeros).
"""
import copy

import numpy as np
import pytest

from conftest import GPU_FAMILIES, load_golden, rel_err
from tinygpt_amd import known_desc, synth
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY

TOL_TINY = 1.5e-2        # floor-sized: 3x the largest oracle-vs-reordered-oracle distance measured on the fixtures in this mode (4.9e-3)
TOL_EXACT = 2e-5        # two fp32 schedules with NO flipped input


def tiny_pair_cpu(fam, dtype):
    from oracle.oracle_ffi import OracleModel
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, dtype, max_batch=g["prompt"].shape[0])
    seed, std = int(g["seed"]), float(g["std"])
    a = OracleModel(d).load_synthetic(seed, std).finalize()
    b = OracleModel(d).load_synthetic(seed, std).finalize()
    b.set_act16(True)
    return a, b, g


@pytest.mark.parametrize("fam", ["llama_tiny", "qwen2_tiny"])
def test_oracle_act16_is_another_contract_in_bf16_and_nothing_in_fp32(fam, oracle_lib):
    a, b, g = tiny_pair_cpu(fam, "bf16")
    a.forward(g["prompt"]); b.forward(g["prompt"])
    e = rel_err(b.logits(rounded=False), a.logits(rounded=False))
    assert 1e-4 < e < 5e-2, e                              # bf16 inputs: 2^-9 relative per element, visible and bounded
    # against HF: the bf16-input contract stays inside the bound the fp32-activation contract is held to (tests/test_oracle_golden.py)
    assert rel_err(b.logits(rounded=False), g["logits_bf16"][:, 0]) < 8e-2
    assert rel_err(b.logits(rounded=False), g["logits_fp32"][:, 0]) < 3e-2
    a, b, g = tiny_pair_cpu(fam, "fp32")
    a.forward(g["prompt"]); b.forward(g["prompt"])
    np.testing.assert_array_equal(a.logits(rounded=False), b.logits(rounded=False))      # fp32 storage: rounding to the storage dtype is the identity


# ---------------------------------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def hip():
    from tinygpt_amd.ffi import product_backend
    return product_backend()


def make_pair(fam, hip, max_batch=1, dtype="bf16", max_ctx=None):
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd.ffi import Model
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, dtype, max_batch=max(max_batch, g["prompt"].shape[0]))
    if max_ctx:
        d.max_ctx = max_ctx
    seed, std = int(g["seed"]), float(g["std"])
    gpu = Model(d, hip).load_synthetic(seed, std).finalize()
    gpu.set_option("act.round16", 1)
    ref = OracleModel(d).load_synthetic(seed, std).finalize()
    ref.set_act16(True)
    return gpu, ref, g, d


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("fam", GPU_FAMILIES)
def test_act16_prefill_and_teacher_forced_steps_vs_oracle(fam, dtype, hip, oracle_lib):
    """every family's fixture: the prompt, then every step forced with the golden ids; logits and cache rows vs the oracle in the same mode"""
    gpu, ref, g, d = make_pair(fam, hip, dtype=dtype)
    assert gpu.get_option("act.round16") == 1
    ids = g["ids_bf16"]
    gpu.forward(g["prompt"]); ref.forward(g["prompt"])
    assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY
    if dtype == "bf16":       # against HF itself: inside the bounds the fp32-activation contract is held to (tests/test_hip_parity.py: 2e-2 / 8e-2), fp32 golden a little wider
        assert rel_err(gpu.logits(rounded=False), g["logits_bf16"][:, 0]) < 8e-2
        assert rel_err(gpu.logits(rounded=False), g["logits_fp32"][:, 0]) < 3e-2
    for i in range(1, ids.shape[1]):
        gpu.forward(ids[:, i - 1:i]); ref.forward(ids[:, i - 1:i])
        assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY, f"step {i}"
    for layer in range(d.layers):
        kg, vg = gpu.read_kv(0, layer); kr, vr = ref.read_kv(0, layer)
        n = ref.past_length
        assert rel_err(kg[:, :n], kr[:, :n]) < 2e-2 and rel_err(vg[:, :n], vr[:, :n]) < 2e-2      # a flipped row is one 16-bit ulp off (2^-8)


@pytest.mark.gpu
@pytest.mark.parametrize("fam", GPU_FAMILIES)
def test_act16_every_linear_rounds_its_input_first_cache_rows_of_layer_two_exact(fam, hip, oracle_lib):
    """Which Linear rounds, checked where no flip has happened yet: the SECOND layer's cache row of the FIRST token is the image of that token's whole
    first layer (qkv, o_proj, gate_up / c_fc, down / c_proj inputs, each rounded) and of its own qkv input, and depends on nothing else.  Without a flipped
    input the 16-bit rows are bit-identical on both sides (a flip: a few per cent per token; once one has happened, every later row carries its 1e-5
    perturbation and re-rounds differently in ~half of the cases — which is why only the FIRST rows are a sharp instrument); a Linear that did not round, or
    rounded elsewhere, would move them in every prompt.  Every prefill path — 1 token + 1 decode step: the GEMV kernels of a step (direct attention with the
    o_proj product in its epilogue at head_dim 64); 3 tokens: passes through the decode kernels, 4 positions at a time; 9 / 40 / 100: skinny MFMA GEMMs (fp32
    rows rounded while staging; stored terms with an all-zero second term; the LDS-DMA ring kernel); 150: tiled GEMMs — over 16 prompts each: at least 4 with the row exact
    (a flip somewhere in ~1500 rounded inputs: measured in a quarter to a half of the prompts); every logits vector within the floor-sized bound.  The lm_head input: the logits of one-token prompts exact in the majority of 12."""
    gpu, ref, g, d = make_pair(fam, hip, max_ctx=64 if fam.startswith("gpt2") else 192)      # (the GPT-2 fixture has 64 learned positions)
    layer = d.layers - 1
    for plen in (1, 3, 9, 40, 100, 150):
        if plen + 2 > d.max_ctx:
            continue
        good = 0
        for k in range(16):
            prompt = synth.synth_prompt(d.vocab, plen, 1000 + 31 * k + plen)[None, :]
            gpu.reset_cache(); ref.reset_cache()
            gpu.forward(prompt); ref.forward(prompt)
            assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY, (plen, k)
            if plen == 1:
                tok = ref.sample(GREEDY); gpu.sample(GREEDY)
                gpu.forward(tok[None, :]); ref.forward(tok[None, :])
                assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY, (plen, k)
            (kg, vg), (kr, vr) = gpu.read_kv(0, layer), ref.read_kv(0, layer)
            good += np.array_equal(kg[0], kr[0]) and np.array_equal(vg[0], vr[0])
        assert good >= 4, (plen, good)          # measured 8-16 of 16; a Linear without its rounding: 0
    hits = 0
    for k in range(12):
        prompt = synth.synth_prompt(d.vocab, 1, 2000 + k)[None, :]
        gpu.reset_cache(); ref.reset_cache()
        gpu.forward(prompt); ref.forward(prompt)
        hits += rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_EXACT
    assert hits >= 7, hits


@pytest.mark.gpu
@pytest.mark.parametrize("fam,rows", [("llama_tiny", 2), ("llama_tiny", 3), ("qwen2_tiny", 4), ("llama_tiny", 5), ("qwen2_tiny", 7), ("mistral_tiny", 20), ("llama_tiny", 40), ("qwen2_tiny", 70)])
def test_act16_batched_steps_vs_oracle(fam, rows, hip, oracle_lib):
    """batch rows share the weight pass (GEMV groups up to 2 rows; skinny MFMA products from 3 rows: fp32 rows rounded ONCE while staging, stored terms with
    an all-zero second term, the LDS-DMA ring kernel from 17 rows): every row vs the oracle in the same mode, teacher-forced through the captured step"""
    gpu, ref, g, d = make_pair(fam, hip, max_batch=rows)
    p = g["prompt"]
    V = d.vocab
    ids = np.concatenate([(p + 3 * b) % V for b in range(rows)])
    gpu.forward(ids); ref.forward(ids)
    assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY
    tok = ref.sample(GREEDY)
    for step in range(5):
        onehot = np.full((rows, V), -1.0, np.float32); onehot[np.arange(rows), tok] = 1.0
        gpu.set_logits(onehot); np.testing.assert_array_equal(gpu.sample(GREEDY), tok)
        tg = gpu.decode(1, GREEDY)[0]
        tr = ref.decode(1, GREEDY)[0]
        lg, lr = gpu.logits(rounded=False), ref.logits(rounded=False)
        assert rel_err(lg, lr) < TOL_TINY, (step, rel_err(lg, lr))
        top2 = np.sort(lr, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 2 * TOL_TINY * np.abs(lr).max()
        np.testing.assert_array_equal(tg[clear], tr[clear])
        tok = tr


def shrunk(name, layers, vocab, max_ctx, dtype="bf16"):
    d = copy.deepcopy(known_desc(name, dtype))
    d.layers, d.vocab, d.max_ctx, d.max_batch = layers, vocab, max_ctx, 1
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("name,plen", [("llama-3.2-1b", 2048), ("llama-3.2-1b", 300), ("qwen2.5-0.5b", 1100), ("mistral-7b-v0.3", 1024)])
def test_act16_long_prompts_on_the_one_term_kernels_vs_oracle(name, plen, hip, oracle_lib):
    """Real layer geometries (2 layers, small vocabulary), prompts long enough for the eight-wave LDS-DMA GEMMs, whose act.round16 forms neither stage nor
    multiply the second term (kernels/gemm_dma.h template LO): prefill logits and four decode steps (split / direct attention + K-sliced or in-launch
    o_proj, all with the rounded o_proj input) vs the oracle in the same mode; then the same prompt with the one-term kernels switched off
    (act.one_term_kernels 0: the two-term kernels read the all-zero second term): BIT-identical — adding exact zeros changes nothing."""
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd.ffi import Model
    d = shrunk(name, 2, 4096, plen + 64)
    gpu = Model(d, hip).load_synthetic(1234, 0.02).finalize()
    gpu.set_option("act.round16", 1)
    ref = OracleModel(d).load_synthetic(1234, 0.02).finalize()
    ref.set_act16(True)
    oracle_lib.set_threads(32)
    try:
        prompt = synth.synth_prompt(d.vocab, plen, 77)[None, :]
        gpu.forward(prompt); ref.forward(prompt)
        lp = gpu.logits(rounded=False).copy()
        assert rel_err(lp, ref.logits(rounded=False)) < TOL_TINY
        tok = ref.sample(GREEDY); gpu.sample(GREEDY)
        for step in range(4):
            gpu.forward(tok[None, :] if tok.ndim == 1 else tok); ref.forward(tok[None, :] if tok.ndim == 1 else tok)
            assert rel_err(gpu.logits(rounded=False), ref.logits(rounded=False)) < TOL_TINY, step
            tok = ref.sample(GREEDY); gpu.sample(GREEDY)
    finally:
        oracle_lib.set_threads(8)
    gpu.set_option("act.one_term_kernels", 0)
    gpu.reset_cache(); gpu.forward(prompt)
    np.testing.assert_array_equal(gpu.logits(rounded=False), lp)
    # and the default contract is another function: the option is not a no-op on this path
    gpu.set_option("act.round16", 0)
    gpu.reset_cache(); gpu.forward(prompt)
    assert rel_err(gpu.logits(rounded=False), lp) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["llama-3.2-1b", "llama-3.2-3b"])
def test_act16_flip_floor_and_full_depth_vs_oracle(name, hip, oracle_lib):
    """Llama-3.2-1B (16 layers) and Llama-3.2-3B (28 layers, head_dim 128) at FULL depth and vocabulary, 192-token prompt + 4 steps: (a) the floor of the mode — the oracle against its own reordered schedule,
    both rounding every Linear input; (b) the HIP path against the oracle, granted 2.5x that floor (and never more than 3e-2).  Printed for
    profiles/r04_act16.txt."""
    from fullsize_util import oracle_trajectory
    from tinygpt_amd.ffi import Model
    PROMPT, STEPS = 192, 4          # the floor of the mode is set by the depth (every Linear input of every layer may flip), not by the prompt length
    a = oracle_trajectory(oracle_lib, name, PROMPT, 3, STEPS, act16=True, kv_layers=())
    b = oracle_trajectory(oracle_lib, name, PROMPT, 3, STEPS, act16=True, reorder=True, forced=a, kv_layers=())
    gpu = Model(a.desc, hip).load_synthetic(1234, 0.02).finalize()
    gpu.set_option("act.round16", 1)
    gpu.forward(a.prompt)
    floor, err = [], []
    for step in range(STEPS + 1):
        la = a.logits[step]
        floor.append(rel_err(b.logits[step], la)); err.append(rel_err(gpu.logits(rounded=False), la))
        if step < STEPS:
            gpu.forward(a.toks[step][None, :])
    print(f"act.round16, {name} full depth: oracle vs reordered oracle", ["%.2e" % e for e in floor], " HIP vs oracle", ["%.2e" % e for e in err])
    # (printed, not asserted from below: the floor of this contract sits well above the fp32-activation contract's 3-5e-4 end to end at this depth)
    assert max(err) < min(3e-2, 2.5 * max(floor)), (floor, err)
