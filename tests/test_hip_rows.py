"""Per-row sequence lifecycle (include/tgx.h ABI 3: tgx_reset_row / tgx_forward_row / tgx_sample_row / tgx_past_length_row) — the kernel-contract half
of the reference's continuous-batching TODO (README.md:33-34; its engine rebuilds the batch per request, src/engine/GPTEngine.cpp:67-84,180-232, and
its KVCacheManager has one pastLength for all rows, src/engine/CacheManager.h:44-51).

A row of a RUNNING batch is retired and another prompt is prefilled into it; the batch then decodes with rows of different lengths (every step kernel
reads the position per row).  Held to:
  * the refilled row == the same prompt run ALONE, within the batch-invariance bound of tests/test_hip_parity.py (1e-3 on the logits, greedy ids equal
    wherever the solo run's top-2 gap exceeds 2e-3): batches of 2 rows ride the GEMV step, 4 / 8 rows the matrix-core step — other kernels, the same math;
  * the OTHER rows are untouched: bit-identical logits and ids to a control batch that was never refilled (a row's arithmetic reads only its own state);
  * a batch can grow by one row (row == batch); the row's length, cache read-back and the error paths (refill without reset, decode before the row
    has a token, tgx_forward on a ragged batch) behave as include/tgx.h says."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, TgxError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from tinygpt_amd.ffi import product_backend
    return product_backend()


def make(fam, hip, max_batch, dtype="bf16", max_ctx=None):
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, dtype, max_batch=max_batch)
    if max_ctx:
        d.max_ctx = max_ctx
    return Model(d, hip).load_synthetic(int(g["seed"]), float(g["std"])).finalize(), g


def force(m, toks):
    """make `toks` [rows] the current tokens of all rows (one-hot logits -> greedy sample), then one decode step"""
    V = m.desc.vocab
    onehot = np.full((len(toks), V), -1.0, np.float32); onehot[np.arange(len(toks)), toks] = 1.0
    m.set_logits(onehot)
    np.testing.assert_array_equal(m.sample(GREEDY), toks)
    return m.decode(1, GREEDY)[0].copy()


def solo_run(fam, hip, prompt, steps, dtype):
    m, _ = make(fam, hip, 1, dtype)
    m.forward(prompt[None, :])
    logits = [m.logits(rounded=False)[0].copy()]
    toks = [int(m.sample(GREEDY)[0])]
    for _ in range(steps):
        toks.append(int(m.decode(1, GREEDY)[0, 0]))
        logits.append(m.logits(rounded=False)[0].copy())
    return toks, logits


def check_row(lb, tok_b, l1, tok_next_1):
    assert rel_err(lb[None, :], l1[None, :]) < 1e-3, rel_err(lb[None, :], l1[None, :])
    top2 = np.sort(l1)[-2:]
    if (top2[1] - top2[0]) > 2e-3 * np.abs(l1).max():
        assert int(tok_b) == int(tok_next_1)


@pytest.mark.parametrize("rows,at", [(2, 1), (4, 2), (8, 0)])
@pytest.mark.parametrize("fam,dtype", [("llama_tiny", "bf16"), ("qwen2_tiny", "bf16"), ("mistral_tiny", "fp16"), ("qwen3_tiny", "bf16")])
def test_a_row_retired_and_refilled_mid_run_equals_its_solo_run(fam, dtype, rows, at, hip):
    STEPS = 5
    gpuB, g = make(fam, hip, rows, dtype)
    ctrl, _ = make(fam, hip, rows, dtype)
    p = g["prompt"][0]
    V = gpuB.desc.vocab
    ids = np.stack([(p + 5 * b + 1) % V for b in range(rows)])
    newp = ((p[:5] * 7 + 3) % V).astype(np.int64)                 # the prompt that takes over the retired row: another length than the live rows'
    solo_toks, solo_logits = solo_run(fam, hip, newp, STEPS, dtype)
    for m in (gpuB, ctrl):
        m.forward(ids); m.sample(GREEDY); m.decode(3, GREEDY)
    L0 = gpuB.past_length
    assert L0 == len(p) + 3 and all(gpuB.past_length_row(r) == L0 for r in range(rows))
    # ---- retire row `at`, prefill the new prompt into it
    gpuB.reset_row(at)
    assert gpuB.past_length_row(at) == 0 and gpuB.past_length == (L0 if rows > 1 else 0)
    with pytest.raises(TgxError) as ei:                            # the row has no current token: the batch cannot step
        gpuB.decode(1, GREEDY)
    assert ei.value.status == 4
    gpuB.forward_row(at, newp)
    assert gpuB.past_length_row(at) == len(newp) and gpuB.past_length == L0
    with pytest.raises(TgxError) as ei:
        gpuB.decode(1, GREEDY)
    assert ei.value.status == 4
    lb = gpuB.logits(rounded=False)
    check_row(lb[at], gpuB.sample_row(at, GREEDY), solo_logits[0], solo_toks[0])
    others = [r for r in range(rows) if r != at]
    np.testing.assert_array_equal(lb[others], ctrl.logits(rounded=False)[others])          # the other rows' logits were not touched
    k_at, _ = gpuB.read_kv(at, 0)
    assert k_at.shape[0] == len(newp)                              # the row's own length, not the batch's longest
    # ---- the batch decodes on with rows of different lengths; the refilled row is forced with its solo run's tokens, the others free-run
    cur_b = gpuB.sample(GREEDY).copy()                             # idempotent for rows whose logits did not change
    cur_c = ctrl.sample(GREEDY).copy()
    np.testing.assert_array_equal(cur_b[others], cur_c[others])
    for step in range(STEPS):
        cur_b[at] = solo_toks[step]
        nxt_b = force(gpuB, cur_b)
        nxt_c = force(ctrl, cur_c)
        lb, lc = gpuB.logits(rounded=False), ctrl.logits(rounded=False)
        check_row(lb[at], nxt_b[at], solo_logits[step + 1], solo_toks[step + 1])
        np.testing.assert_array_equal(lb[others], lc[others])      # bit-identical: a row's arithmetic reads only its own state
        np.testing.assert_array_equal(nxt_b[others], nxt_c[others])
        cur_b, cur_c = nxt_b, nxt_c
        assert gpuB.past_length_row(at) == len(newp) + step + 1 and gpuB.past_length == L0 + step + 1
    # ---- misuse is loud
    with pytest.raises(TgxError) as ei:
        gpuB.forward_row(at, newp)                                 # refill without reset
    assert ei.value.status == 4
    with pytest.raises(TgxError) as ei:
        gpuB.forward(np.stack([newp[:1]] * rows))                  # tgx_forward is the whole-batch call: not on a ragged batch
    assert ei.value.status == 4
    with pytest.raises(TgxError) as ei:
        gpuB.reset_row(rows)
    assert ei.value.status == 1
    gpuB.reset_cache()
    assert gpuB.past_length == 0 and gpuB.past_length_row(at) == 0
    gpuB.forward(ids); ctrl.reset_cache(); ctrl.forward(ids)      # and the whole-batch path is back to normal afterwards
    np.testing.assert_array_equal(gpuB.logits(rounded=False), ctrl.logits(rounded=False))


@pytest.mark.parametrize("fam,plen", [("llama_tiny", 40), ("qwen2_tiny", 150), ("gpt2_hd64", 6)])
def test_a_batch_grows_by_one_row_and_long_prompts_refill_through_the_matrix_core_prefill(fam, plen, hip):
    """row == batch appends a sequence to a live batch; prompts of 40 / 150 tokens take the skinny / tiled matrix-core prefill as a one-row pass
    (GPT-2: passes through the decode kernels with learned positions from 0)"""
    STEPS = 4
    gpuB, g = make(fam, hip, 4, max_ctx=64 if fam.startswith("gpt2") else 192)
    p = g["prompt"][0]
    V = gpuB.desc.vocab
    ids = np.stack([(p + 11 * b + 2) % V for b in range(2)])
    gpuB.forward(ids); gpuB.sample(GREEDY); gpuB.decode(2, GREEDY)
    newp = ((np.arange(plen) * 37 + 5) % V).astype(np.int64)
    cfg, _ = load_golden(fam)
    d1 = desc_from_hf_config(cfg, "bf16", max_batch=1); d1.max_ctx = gpuB.desc.max_ctx
    solo = Model(d1, hip).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    solo.forward(newp[None, :])
    l1 = [solo.logits(rounded=False)[0].copy()]; t1 = [int(solo.sample(GREEDY)[0])]
    for _ in range(STEPS):
        t1.append(int(solo.decode(1, GREEDY)[0, 0])); l1.append(solo.logits(rounded=False)[0].copy())
    with pytest.raises(TgxError):
        gpuB.forward_row(3, newp)                                   # rows are appended in order: 2 is the next free one
    gpuB.forward_row(2, newp)
    assert gpuB.batch == 3 and gpuB.past_length_row(2) == plen and gpuB.past_length == max(plen, len(p) + 2)
    check_row(gpuB.logits(rounded=False)[2], gpuB.sample_row(2, GREEDY), l1[0], t1[0])
    cur = gpuB.sample(GREEDY).copy()
    for step in range(STEPS):
        cur[2] = t1[step]
        nxt = force(gpuB, cur)
        check_row(gpuB.logits(rounded=False)[2], nxt[2], l1[step + 1], t1[step + 1])
        cur = nxt
