"""GPU Sampler::sample (kernels/sampler.h) against the golden sampler vectors and against the oracle's draw."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import SamplerCfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(oracle_lib):
    """A tiny model whose vocabulary matches the sampler vectors (V = 512); only its logits buffer is used."""
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd.ffi import Model, product_backend
    cfg, g = load_golden("llama_tiny")
    cfg = dict(cfg, vocab_size=512)
    d = desc_from_hf_config(cfg, "bf16")
    gpu = Model(d, product_backend()).load_synthetic(7, 0.05).finalize()
    ref = OracleModel(d).load_synthetic(7, 0.05).finalize()
    return gpu, ref


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLDEN, "sampler", "golden.npz"))


def test_kept_set_and_probs_match_golden(pair, vec):
    gpu, _ = pair
    for li in range(int(vec["n_logits"])):
        logits = vec[f"logits{li}"]
        for ci in range(int(vec["n_cfgs"])):
            T, K, P, M = vec[f"case{li}_{ci}_cfg"]
            cfg = SamplerCfg(float(T), int(K), float(P), float(M))
            gpu.set_logits(logits)
            gpu.sample(cfg, seed=123)
            probs = gpu.probs()[0]
            want = vec[f"case{li}_{ci}_probs"]
            if len(np.unique(logits)) == logits.size or K == 0:
                np.testing.assert_array_equal(probs > 0, want > 0, err_msg=f"kept set differs: logits{li} cfg{ci}")
                np.testing.assert_allclose(probs, want, rtol=3e-5, atol=1e-8)
            elif len(np.unique(logits)) == 1:
                # all-equal logits: the top-p cut lands exactly on cum == top_p (45 * 1/50 vs 0.9), which float
                # cumsum order decides; the kept COUNT may differ by one, the distribution must stay uniform
                n, n_want = int((probs > 0).sum()), int((want > 0).sum())
                assert abs(n - n_want) <= 1 and n >= 1
                np.testing.assert_allclose(probs[probs > 0], 1.0 / n, rtol=1e-5)
            else:   # ties at the k-th value: which tied index survives torch.topk is implementation-defined
                np.testing.assert_allclose(np.sort(probs), np.sort(want), rtol=3e-5, atol=1e-8)


def test_draws_match_oracle(pair, vec):
    """Same logits, config and seed -> the same token as the oracle's inverse-CDF draw, and a kept token."""
    gpu, ref = pair
    cfgs = [SamplerCfg(0.8, 0, 0.9, 0.0), SamplerCfg(0.7, 50, 1.0, 0.0), SamplerCfg(1.0, 0, 1.0, 0.05),
            SamplerCfg(0.8, 50, 0.9, 0.05), SamplerCfg(0.0, 0, 0.5, 0.0)]
    logits = vec["logits0"]
    seen = set()
    for cfg in cfgs:
        for seed in range(12):
            gpu.set_logits(logits); ref.be.set_logits(ref._ctx, np.ascontiguousarray(logits[None]).ctypes.data_as(
                __import__("ctypes").POINTER(__import__("ctypes").c_float)), 1); ref.batch = 1
            a = int(gpu.sample(cfg, seed=seed)[0]); b = int(ref.sample(cfg, seed=seed)[0])
            assert a == b, (cfg.temperature, cfg.top_k, cfg.top_p, cfg.min_p, seed)
            assert gpu.probs()[0][a] > 0
            seen.add(a)
    assert len(seen) >= 3          # the draw really varies with the seed (the distribution is peaked)


def test_greedy_tie_break_lowest_index(pair, vec):
    gpu, _ = pair
    gpu.set_logits(vec["logits1"])                       # duplicated maximum at indices 10 and 200
    assert int(gpu.sample(SamplerCfg())[0]) == int(vec["argmax1"]) == 10
    gpu.set_logits(vec["logits2"])                       # all equal
    assert int(gpu.sample(SamplerCfg())[0]) == 0


@pytest.mark.parametrize("fam", ["qwen2_tiny", "gpt2_hd64"])
def test_sampled_decode_matches_oracle(fam, oracle_lib):
    """Whole sampled decode loop (forward -> sample -> embed) with T=0.8/top-p 0.9, the CLI defaults
    (examples/inference/main.cpp:36-37): same seed -> same ids as the oracle.  GPT-2: the pick kernel also adds the learned
    position row of the NEXT position to the sampled token's embedding; its fixture prompt is the CLI's batch of 4."""
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd.ffi import Model, product_backend
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, "bf16", max_batch=g["prompt"].shape[0])
    gpu = Model(d, product_backend()).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    ref = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    sc = SamplerCfg(0.8, 0, 0.9, 0.0)
    gpu.forward(g["prompt"]); ref.forward(g["prompt"])
    np.testing.assert_array_equal(gpu.sample(sc, seed=42), ref.sample(sc, seed=42))
    np.testing.assert_array_equal(gpu.decode(12, sc, seed=42), ref.decode(12, sc, seed=42))


@pytest.mark.parametrize("V", [5003, 70001, 151936])
def test_staged_sampler_many_workgroups_vs_oracle(V, oracle_lib):
    """Vocabularies that span several workgroups (4096 entries each... 1024 per workgroup), are not a multiple of 4 (the second
    batch row starts unaligned) and need both index digit levels (V > 2048): kept set, probabilities and draws of a
    2-row batch against the oracle's sort-based Sampler restatement, for every filter combination."""
    import ctypes
    from oracle.oracle_ffi import OracleModel, filter_logits
    from tinygpt_amd.ffi import Model, product_backend
    cfg, g = load_golden("llama_tiny")
    cfg = dict(cfg, vocab_size=V, hidden_size=64, intermediate_size=64, num_attention_heads=1, num_key_value_heads=1, num_hidden_layers=1)
    d = desc_from_hf_config(cfg, "bf16", max_batch=2)
    gpu = Model(d, product_backend()).load_synthetic(3, 0.05).finalize()
    ref = OracleModel(d).load_synthetic(3, 0.05).finalize()
    rng = np.random.default_rng(V)
    peaked = (rng.standard_normal((2, V)) * 2.5).astype(np.float32)         # realistic: a few dominant tokens
    flat = (rng.standard_normal((2, V)) * 0.05).astype(np.float32)          # nearly uniform: top-p keeps ~90 % of the entries
    tied = np.round(peaked * 2) / 2                                         # heavy ties: the index digits decide
    const = np.full((2, V), 0.25, np.float32)                               # every logit equal: the compacted list of a filter is the whole vocabulary (round 6: the tail streams it)
    cfgs = [SamplerCfg(0.8, 0, 0.9, 0.0), SamplerCfg(0.7, 50, 1.0, 0.0), SamplerCfg(1.0, 0, 1.0, 0.05), SamplerCfg(0.8, 50, 0.9, 0.05),
            SamplerCfg(0.0, 0, 0.5, 0.0), SamplerCfg(1.0, 0, 1.0, 0.0), SamplerCfg(1.3, 3000, 0.97, 0.0), SamplerCfg(0.9, V + 10, 0.999, 0.0)]
    for logits in (peaked, flat, tied, const):
        for sc in cfgs:
            gpu.set_logits(logits)
            ref.be.set_logits(ref._ctx, np.ascontiguousarray(logits).ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 2); ref.batch = 2
            for seed in (1, 2, 3):
                a = gpu.sample(sc, seed=seed); b = ref.sample(sc, seed=seed)
                probs = gpu.probs()
                for row in range(2):
                    _, want = filter_logits(sc, logits[row])
                    kept, kept_w = probs[row] > 0, want > 0
                    # the cut of top-p sits where the cumulative mass crosses P: the oracle accumulates it in float like torch.cumsum
                    # (error up to ~V * 2^-24 of the total), the GPU in exact 2^-40 fixed point — they may disagree on the entries
                    # whose mass lies inside that error band around the crossing, and on nothing else
                    diff = kept != kept_w
                    if sc.top_p >= 1:
                        assert not diff.any(), (V, sc.temperature, sc.top_k, sc.top_p, sc.min_p, row)
                    else:
                        band = float(np.maximum(want, probs[row])[diff].sum())
                        # (all-equal logits: k equal masses can put the cumulative sum EXACTLY on top_p — 45 x 1/50 vs 0.9 — where float and fixed point may
                        #  each land on either side: one entry's mass of slack there, as in test_kept_set_and_probs_match_golden)
                        slack = float(np.maximum(want, probs[row]).max()) if logits is const else 0.0
                        assert band <= 4 * V * 2.0 ** -24 + slack * (1 + 1e-5), (V, sc.temperature, sc.top_k, sc.top_p, sc.min_p, row, int(diff.sum()), band)
                    if (kept == kept_w).all():
                        np.testing.assert_allclose(probs[row], want, rtol=5e-5, atol=1e-9)
                        assert int(a[row]) == int(b[row])
                    assert probs[row][int(a[row])] > 0
