"""Paged KV cache (round 6; option kv.budget_tokens before tgx_finalize; include/tgx.h) — the kernel-contract half of the reference's "Paged Attention" TODO
(README.md:32-34).  The reference's KVCacheManager grows every row's cache by concat (src/engine/CacheManager.h:24-51); the unpaged layout gives every row a
max_ctx slab.  Paged: per-layer pools of 128-token blocks shared by the rows, a block table per row on the device, blocks assigned as a sequence grows and
returned when it is retired.  Held to:
  * the SAME results as the unpaged cache — bit for bit on the same option set (every decode attention form has a paged instantiation: the unpaged kernel
    with another address computation): prompts through the decode-kernel passes, decode steps on the direct (with and without the o_proj strip), split and
    matrix-core attention forms, the batched step with VALU / matrix-core attention with and without the QKV finish in the prologue, across block boundaries,
    four families / both head sizes / a QKV bias / Qwen3's q-k norm;
  * rows of very different lengths whose summed length exceeds max_ctx — impossible with one slab of the same total size — decode together and each equals
    its run in an unpaged batch;
  * the budget: a row that needs a block when none is free is refused (TGX_ERR_CONTEXT), retiring another row frees its blocks, and the refused row then
    proceeds; kv.free_tokens accounts for every block; tgx_read_kv / tgx_write_kv go through the table."""
import numpy as np
import pytest

from conftest import load_golden
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, TgxError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from tinygpt_amd.ffi import product_backend
    return product_backend()


def make(fam, hip, dtype="bf16", max_batch=1, max_ctx=512, budget=0, opts=()):
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, dtype, max_batch=max_batch)
    d.max_ctx = max_ctx
    m = Model(d, hip)
    if budget:
        m.set_option("kv.budget_tokens", budget)
    m.load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    for k, v in tuple(opts):
        m.set_option(k, v)
    return m, g


# Every decode attention form has a paged instantiation, so a paged context takes the forms an unpaged one takes: the comparisons below run both sides on the
# SAME option set.  Prompts go through the decode-kernel passes here (prefill.mfma 0: the matrix-core prefill picks other — equivalent, not bit-identical —
# forms for a paged cache; test_matrix_core_prefill_into_a_paged_cache holds that path).
FORMS = {
    "default": (("prefill.mfma", 0),),                                                     # direct attention with the o_proj strip in its launch
    "split": (("prefill.mfma", 0), ("attn.direct_max", 0)),                                 # split attention, merged by the K-sliced o_proj
    "plain_direct": (("prefill.mfma", 0), ("oproj.fused", 0)),                              # the plain direct form, o_proj as its own launch
    "plain_split": (("prefill.mfma", 0), ("oproj.fused", 0), ("attn.direct_max", 0)),       # split + combine launch
    "mfma_long": (("prefill.mfma", 0), ("attn.direct_max", 0), ("attn.mfma_min", 64)),      # the long-context matrix-core form (from 64 keys here)
}


@pytest.mark.parametrize("forms", list(FORMS))
@pytest.mark.parametrize("fam,dtype", [("llama_tiny", "bf16"), ("qwen2_tiny", "bf16"), ("mistral_tiny", "fp16"), ("qwen3_tiny", "bf16")])
def test_paged_equals_unpaged_bit_for_bit(fam, dtype, forms, hip):
    opts = FORMS[forms]
    paged, g = make(fam, hip, dtype, max_ctx=512, budget=512, opts=opts)
    plain, _ = make(fam, hip, dtype, max_ctx=512, opts=opts)
    assert paged.get_option("kv.free_tokens") == 512 and plain.get_option("kv.free_tokens") == -1
    V = paged.desc.vocab
    p = g["prompt"][0]
    prompt = np.concatenate([(p * (3 + i) + i) % V for i in range(14)])[:121].astype(np.int64)      # 121 tokens: the decode crosses the first block boundary (128)
    for m in (paged, plain):
        m.forward(prompt[None, :])
    np.testing.assert_array_equal(paged.logits(rounded=False), plain.logits(rounded=False))
    assert paged.get_option("kv.free_tokens") == 512 - 128
    ta, tb = paged.sample(GREEDY), plain.sample(GREEDY)
    np.testing.assert_array_equal(ta, tb)
    da, db = paged.decode(150, GREEDY), plain.decode(150, GREEDY)              # 16-step graphs, two more blocks on the way
    np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(paged.logits(rounded=False), plain.logits(rounded=False))
    assert paged.past_length == 271 and paged.get_option("kv.free_tokens") == 512 - 3 * 128
    for layer in (0, paged.desc.layers - 1):
        for a, b in zip(paged.read_kv(0, layer), plain.read_kv(0, layer)):
            np.testing.assert_array_equal(a, b)
    # write_kv through the table: a round trip of perturbed rows reads back, and the next step sees them on both sides alike
    k, v = plain.read_kv(0, 0)
    k2 = (k * 0.5).astype(np.float32)
    for m in (paged, plain):
        m.write_kv(0, 0, k2, v)
    np.testing.assert_array_equal(paged.read_kv(0, 0)[0], plain.read_kv(0, 0)[0])
    np.testing.assert_array_equal(paged.decode(3, GREEDY), plain.decode(3, GREEDY))
    np.testing.assert_array_equal(paged.logits(rounded=False), plain.logits(rounded=False))
    # reset returns every block
    paged.reset_cache()
    assert paged.get_option("kv.free_tokens") == 512
    paged.forward(prompt[None, :5]); plain.reset_cache(); plain.forward(prompt[None, :5])
    np.testing.assert_array_equal(paged.logits(rounded=False), plain.logits(rounded=False))


STEPS = {
    "gemv": (("prefill.mfma", 0), ("decode.mfma_min_batch", 1 << 20)),                      # the GEMV step, rows in groups of four
    "mfma_default": (("prefill.mfma", 0),),                                                 # the matrix-core step as four rows take it
    "mfma_valu_plain": (("prefill.mfma", 0), ("attn.raw_fuse", 0), ("attn.batch_mfma", 0)),  # the QKV finish as its own launch, VALU attention
    "mfma_valu_raw": (("prefill.mfma", 0), ("attn.raw_fuse", 2), ("attn.batch_mfma", 0)),   # VALU attention that finishes the QKV rows in its prologue
    "mfma_attn": (("prefill.mfma", 0), ("attn.batch_mfma", 1)),                              # matrix-core attention (as from 17 rows) + the QKV finish in its prologue
    "mfma_attn_plain": (("prefill.mfma", 0), ("attn.batch_mfma", 1), ("attn.raw_fuse", 0)),
}


@pytest.mark.parametrize("step", list(STEPS))
@pytest.mark.parametrize("fam", ["llama_tiny", "mistral_tiny", "qwen3_tiny"])
def test_rows_of_very_different_lengths_share_a_budget_smaller_than_their_slabs(fam, step, hip):
    """max_ctx 384, four rows: unpaged that is 4 x 384 = 1536 tokens of cache; the paged context gets 896 (seven blocks).  Rows of 300 + 200 + 60 + 20 prompt tokens (580 > max_ctx)
    decode 25 steps together; each row equals the same row of an unpaged batch run on the same kernels, bit for bit."""
    lens = [300, 200, 60, 20]
    paged, g = make(fam, hip, max_batch=4, max_ctx=384, budget=896, opts=STEPS[step])
    plain, _ = make(fam, hip, max_batch=4, max_ctx=384, opts=STEPS[step])
    V = paged.desc.vocab
    p = g["prompt"][0]
    prompts = [np.concatenate([(p * (5 + r + i) + i) % V for i in range(40)])[:n].astype(np.int64) for r, n in enumerate(lens)]
    for m in (paged, plain):
        for r, pr in enumerate(prompts):
            m.forward_row(r, pr)
            m.sample_row(r, GREEDY)
    used = sum((n + 127) // 128 for n in lens) * 128
    assert paged.get_option("kv.free_tokens") == 896 - used == 0
    da, db = paged.decode(25, GREEDY), plain.decode(25, GREEDY)
    np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(paged.logits(rounded=False), plain.logits(rounded=False))
    for r, n in enumerate(lens):
        assert paged.past_length_row(r) == n + 25
        for a, b in zip(paged.read_kv(r, 1), plain.read_kv(r, 1)):
            np.testing.assert_array_equal(a, b)


def test_the_budget_is_enforced_and_retired_rows_return_their_blocks(hip):
    same = STEPS["gemv"] + (("oproj.fused", 0),)                                   # batch and solo on the same kernel path: their ids are compared below
    paged, g = make("llama_tiny", hip, max_batch=3, max_ctx=512, budget=512, opts=same)       # four blocks
    solo, _ = make("llama_tiny", hip, max_batch=1, max_ctx=512, opts=same)
    V = paged.desc.vocab
    p = g["prompt"][0]
    long_p = np.concatenate([(p * (2 + i) + i) % V for i in range(40)])[:250].astype(np.int64)     # two blocks
    mid_p = long_p[:130][::-1].copy()                                                               # two blocks
    paged.forward_row(0, long_p); paged.sample_row(0, GREEDY)
    paged.forward_row(1, mid_p); paged.sample_row(1, GREEDY)
    assert paged.get_option("kv.free_tokens") == 0
    with pytest.raises(TgxError) as ei:                        # a third sequence finds no block
        paged.forward_row(2, long_p[:10])
    assert ei.value.status == 8 and "budget" in str(ei.value)
    assert paged.past_length_row(2) == 0
    paged.decode(5, GREEDY)                                    # rows 0 and 1 still fit their blocks (255, 135)
    with pytest.raises(TgxError) as ei:                        # row 0 would cross into a third block: refused, nothing was stepped
        paged.decode(2, GREEDY)
    assert ei.value.status == 8
    assert paged.past_length_row(0) == 255 and paged.past_length_row(1) == 135
    paged.reset_row(1)                                         # retire the shorter sequence: two blocks come back
    assert paged.get_option("kv.free_tokens") == 256
    ids = paged.decode(6, GREEDY)[:, 0]                        # row 0 crosses the boundary now; the retired row rides along on the scratch block
    paged.forward_row(1, long_p[:10]); first = paged.sample_row(1, GREEDY)
    tail = paged.decode(4, GREEDY)
    # row 0 throughout == the same sequence alone in an unpaged context (the batch invariance bound of tests/test_hip_rows.py: other kernel paths)
    solo.forward(long_p[None, :]); solo.sample(GREEDY)
    want = solo.decode(15, GREEDY)[:, 0]
    np.testing.assert_array_equal(np.concatenate([ids, tail[:, 0]]), want[5:15])
    solo.reset_cache(); solo.forward(long_p[None, :10])
    assert int(solo.sample(GREEDY)[0]) == int(first)
    np.testing.assert_array_equal(solo.decode(4, GREEDY)[:, 0], tail[:, 1])


def test_paged_needs_16_bit_storage_and_is_set_before_finalize(hip):
    cfg, g = load_golden("llama_tiny")
    d = desc_from_hf_config(cfg, "fp32")
    m = Model(d, hip)
    m.set_option("kv.budget_tokens", 256)
    m.load_synthetic(int(g["seed"]), float(g["std"]))
    with pytest.raises(TgxError) as ei:
        m.finalize()
    assert ei.value.status == 2
    m2, _ = make("llama_tiny", hip)
    with pytest.raises(TgxError) as ei:
        m2.set_option("kv.budget_tokens", 256)
    assert ei.value.status == 4


@pytest.mark.parametrize("name,S,dtype", [("llama-3.2-1b", 40, "bf16"), ("llama-3.2-1b", 300, "bf16"), ("llama-3.2-1b", 1100, "bf16"), ("llama-3.2-1b", 3500, "bf16"), ("llama-3.2-1b", 6500, "bf16"),
                                          ("mistral-7b-v0.3", 200, "fp16"), ("qwen2.5-0.5b", 700, "bf16"), ("qwen3-1.7b", 150, "bf16")])
def test_matrix_core_prefill_into_a_paged_cache(name, S, dtype, hip):
    """Prompts of 40 .. 6500 rows (the last: the 20 decode steps run the long-context matrix-core attention) at real layer shapes (2 layers) through the skinny and the tiled matrix-core prefill of a paged context: RoPE + cache append
    (the QKV GEMM's epilogue at head_dim 64, else rope_kv_split_kernel) and the causal prompt attention (attn_prefill_kernel<.., PAGED> plain / key-split /
    lean, attn_prefill_dma_kernel<.., PAGED> from ~3k tokens: the table slice cached in LDS) through the block table.  Both contexts choose their forms by the
    same rules, so the paged one is held to the unpaged one bit for bit: logits, cache rows, greedy continuation."""
    import copy
    from tinygpt_amd import known_desc, synth
    out = []
    BUDGET = ((S + 64 + 127) // 128 + 2) * 128
    for budget in (0, BUDGET):
        d = copy.deepcopy(known_desc(name, dtype))
        d.layers, d.vocab, d.max_ctx = 2, 4096, S + 64
        m = Model(d, hip)
        if budget:
            m.set_option("kv.budget_tokens", budget)
        m.load_synthetic(1234, 0.02).finalize()
        m.forward(synth.synth_prompt(d.vocab, S, 91)[None, :])
        lg = m.logits(rounded=False).copy()
        first = m.sample(GREEDY).copy()
        toks = m.decode(20, GREEDY).copy()
        out.append((lg, first, toks, m.read_kv(0, 0), m.read_kv(0, 1)))
        if budget:
            assert m.get_option("kv.free_tokens") == BUDGET - ((S + 20 + 127) // 128) * 128
        m.close()
    (la, fa, ta, k0a, k1a), (lb, fb, tb, k0b, k1b) = out
    np.testing.assert_array_equal(la, lb)
    np.testing.assert_array_equal(fa, fb)
    np.testing.assert_array_equal(ta, tb)
    for x, y in zip(k0a + k1a, k0b + k1b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("fam,seed", [("llama_tiny", 1), ("llama_tiny", 2), ("mistral_tiny", 3), ("qwen3_tiny", 4)])
def test_serving_churn_under_a_budget_equals_the_unpaged_batch(fam, seed, hip):
    """A serving loop as a host would run it: six rows, sequences of random lengths admitted against a token budget (reserved up
    front from kv.free_tokens' arithmetic), decoded together in calls of random length, retired and replaced — 60 ticks.  The paged context (1280 tokens for
    rows that would own 6 x 384 = 2304 as slabs) and an unpaged one execute the same calls: every live row's tokens are equal call by call, the free-block
    count follows ceil(length / 128) per live row exactly at every tick, and every block is back at the end."""
    B, CTX, BUDGET = 6, 384, 1280
    paged, g = make(fam, hip, max_batch=B, max_ctx=CTX, budget=BUDGET)
    plain, _ = make(fam, hip, max_batch=B, max_ctx=CTX)
    V = paged.desc.vocab
    rng = np.random.default_rng(seed)
    blocks = lambda n: (n + 127) // 128
    length = [0] * B          # tokens in the row's cache; 0 = idle
    target = [0] * B          # its final length
    served = 0
    # the batch is born by a whole-batch call (rows 1..5 retired at once, so that each starts idle)
    first = rng.integers(0, V, size=(B, 3)).astype(np.int64)
    for m in (paged, plain):
        m.forward(first); m.sample(GREEDY)
        for r in range(B):
            m.reset_row(r)
    for tick in range(60):
        reserved = sum(blocks(t) for t in target) * 128
        for r in range(B):
            if length[r] == 0 and rng.random() < 0.6:
                L, new = int(rng.integers(2, 180)), int(rng.integers(1, 120))
                if L + new > CTX or reserved + blocks(L + new) * 128 > BUDGET:
                    continue                                   # admission control: the sequence waits
                prompt = rng.integers(0, V, size=L).astype(np.int64)
                ta = []
                for m in (paged, plain):
                    m.forward_row(r, prompt); ta.append(int(m.sample_row(r, GREEDY)))
                assert ta[0] == ta[1], (tick, r)
                length[r], target[r] = L, L + new
                reserved += blocks(L + new) * 128
        live = [r for r in range(B) if length[r]]
        if live:
            n = int(min(rng.integers(1, 24), min(target[r] - length[r] for r in live)))
            if n > 0:
                da, db = paged.decode(n, GREEDY), plain.decode(n, GREEDY)
                np.testing.assert_array_equal(da[:, live], db[:, live], err_msg=f"tick {tick}")
                for r in live:
                    length[r] += n
        assert paged.get_option("kv.free_tokens") == BUDGET - sum(blocks(x) for x in length) * 128, tick
        for r in live:
            assert paged.past_length_row(r) == plain.past_length_row(r) == length[r]
            if length[r] >= target[r]:
                for m in (paged, plain):
                    m.reset_row(r)
                length[r] = target[r] = 0
                served += 1
    assert served >= 8                                         # the loop really turned sequences over
    for r in range(B):
        paged.reset_row(r)
    assert paged.get_option("kv.free_tokens") == BUDGET


@pytest.mark.parametrize("name,dtype,B", [("llama-3.2-1b", "bf16", 20), ("mistral-7b-v0.3", "fp16", 18), ("qwen2.5-0.5b", "bf16", 24), ("qwen3-1.7b", "bf16", 6)])
def test_batched_steps_at_real_layer_shapes_paged_equals_unpaged(name, dtype, B, hip):
    """Real layer shapes (2 layers), batches that take the matrix-core step with its default forms (from 17 rows: matrix-core attention with the QKV finish in its
    prologue; 6 rows: the VALU direct forms), ragged prompt lengths through the per-row lifecycle, 40 joint steps across a block boundary: token for token and
    logit for logit the unpaged batch."""
    import copy
    from tinygpt_amd import known_desc, synth
    out = []
    lens = [40 + 37 * (r % 5) + r for r in range(B)]                    # 40 .. ~210 tokens: rows at different positions, some crossing 128 / 256 during the steps
    for budget in (0, B * 384):
        d = copy.deepcopy(known_desc(name, dtype))
        d.layers, d.vocab, d.max_ctx, d.max_batch = 2, 4096, 512, B
        m = Model(d, hip)
        if budget:
            m.set_option("kv.budget_tokens", budget)
        m.load_synthetic(1234, 0.02).finalize()
        m.forward(np.zeros((B, 1), dtype=np.int64)); m.sample(GREEDY)
        for r in range(B):
            m.reset_row(r)
            m.forward_row(r, synth.synth_prompt(d.vocab, lens[r], 300 + r)); m.sample_row(r, GREEDY)
        toks = m.decode(40, GREEDY).copy()
        out.append((toks, m.logits(rounded=False).copy(), m.read_kv(B - 1, 1)))
        m.close()
    (ta, la, ka), (tb, lb, kb) = out
    np.testing.assert_array_equal(ta, tb)
    np.testing.assert_array_equal(la, lb)
    for x, y in zip(ka, kb):
        np.testing.assert_array_equal(x, y)
