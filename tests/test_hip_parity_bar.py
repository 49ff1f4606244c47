"""north_star's parity bar — "outputs matching the reference CPU path within 1e-3 relative fp32 (greedy tokens identical)" — held at
the benchmark's own operating point and at full depth on the other BASELINE geometries (VERDICT r2 "next round" #2).

Two comparisons per decode step, both teacher-forced with the ORACLE's token:
  end to end      the GPU's own KV cache (its prefill, its appended rows): <= 1e-3 at the headline point (measured 3.4-5.4e-4), <= 3-4e-3 at 28-32 layers.  Both sides round K / V to bf16 once; where the two fp32
                  values straddle a rounding boundary the stored entries differ by one ulp, and 16-32 layers of such flips put the floor of this
                  comparison at 0.7-1.1e-3 (measured, round 2) — it cannot tell a 5e-4 kernel error from the floor.
  injected cache  the CPU path's cache rows are written into the GPU cache (tgx_write_kv) before the step: the step's inputs are then
                  bit-identical on both sides and the logits are held to <= 1e-4 at the headline point (measured 5e-6 .. 3e-5; <= 5e-4 on the
                  28 / 32-layer models, where the step's own appended row still rounds on the GPU) — the step kernels themselves
                  (RMSNorm, QKV + RoPE + append, split attention + combine, o_proj, gate_up + siluMul, down, lm_head) against the CPU path.
Reference: Attention.h:71-112, CacheManager.h:24-51, GatedMLP.h:37-41, GPTModel.h:51-58."""

import numpy as np
import pytest

from conftest import rel_err
from fullsize_util import BENCH_EVERY, BENCH_LAST, BENCH_S, bench_range_trajectory, oracle_trajectory
from tinygpt_amd import synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend

pytestmark = pytest.mark.gpu


def inject_cache(gpu, traj, n_rows):
    """the oracle's cache rows [0, n_rows) of every layer -> the GPU cache (bf16 values: the conversion is exact)"""
    for layer in range(traj.desc.layers):
        k, v = traj.kv_prefix(layer, n_rows)
        gpu.write_kv(0, layer, k, v)


def force_graph_step(m, tok, vocab):
    """one teacher-forced step through the captured decode graph: make `tok` the current token, replay one step"""
    onehot = np.full((1, vocab), -1.0, np.float32); onehot[0, int(tok[0])] = 1.0
    m.set_logits(onehot)
    assert int(m.sample(GREEDY)[0]) == int(tok[0])
    m.decode(1, GREEDY)


def load_gpus(d, n_gpu=2):
    gpus = [Model(d, product_backend()) for _ in range(n_gpu)]
    for name, bits in synth.synth_checkpoint(d, 1234, 0.02):
        for g in gpus:
            g.upload(name, bits)
    for g in gpus:
        g.finalize()
    return gpus


def test_llama_3_2_1b_bench_range_end_to_end_and_with_the_cpu_paths_cache(oracle_lib):
    """bench.py's workload: 2048-token prompt (MFMA prefill), then the bench's whole decode range — context 2049 .. 2320 — teacher-forced
    through the captured decode graph (what the bench times).  Checked at the first 8 steps and then every 32nd: end to end <= 1e-3 on one
    context, <= 1e-4 on a second one whose cache holds the CPU path's rows; greedy id equal wherever the oracle's top-2 gap exceeds the bound."""
    S, LAST, EVERY = BENCH_S, BENCH_LAST, BENCH_EVERY
    traj = bench_range_trajectory(oracle_lib)
    d = traj.desc
    end2end, injected = load_gpus(d)
    for m in (end2end, injected):
        m.forward(traj.prompt)                                         # bench.py's rank-0 prompt
    e2e, inj = [], []
    for step in range(LAST + 1):
        checked = step < 8 or step % EVERY == 0
        if checked:
            lr = traj.logits[step]
            le, li = end2end.logits(rounded=False), injected.logits(rounded=False)
            e2e.append(rel_err(le, lr)); inj.append(rel_err(li, lr))
            assert e2e[-1] < 1e-3, (step, e2e)                     # north_star's bar, end to end (measured 3.4e-4 .. 5.4e-4 over the range)
            if step:                                                # step 0 = the prefill's logits: no step kernel has run on the injected cache yet
                assert inj[-1] < 1e-4, (step, inj)
            top2 = np.sort(lr[0])[-2:]
            gap = (top2[1] - top2[0]) / np.abs(lr).max()
            if gap > 2e-3:
                assert int(np.argmax(le[0])) == int(np.argmax(lr[0]))
            if gap > 2e-4 and step:
                assert int(np.argmax(li[0])) == int(np.argmax(lr[0]))
        tok = traj.toks[step]
        if step == LAST:
            break
        next_checked = (step + 1) < 8 or (step + 1) % EVERY == 0
        if next_checked:
            inject_cache(injected, traj, S + step)                  # rows [0, pastLength): the step's K / V inputs are the CPU path's
        force_graph_step(end2end, tok, d.vocab)
        force_graph_step(injected, tok, d.vocab)
        assert end2end.past_length == injected.past_length == S + step + 1
    print("bench range, end to end:", ["%.1e" % e for e in e2e])
    print("bench range, injected  :", ["%.1e" % e for e in inj])


@pytest.mark.parametrize("name", ["llama-3.2-3b", "mistral-7b-v0.3"])
def test_full_depth_vs_oracle(name, oracle_lib):
    """BASELINE configs #4 / #5 at FULL depth and vocabulary (3.2 B / 7.2 B parameters; head_dim 128, 24 / 32 query heads over 8 kv heads,
    untied lm_head on Mistral) against the CPU path: 320-token prompt (batched prefill), 4 forced decode steps as tgx_forward passes and 2 more
    through the decode graph; end to end <= 3e-3 (<= 4e-3 over Mistral's 32 layers: its schedule-vs-schedule flip floor alone is 2.1e-3,
    test_other_baseline_geometries_decode_properties), with the CPU path's cache rows <= 5e-4 (the step's own row still rounds on the GPU).
    ModelLlama.h:35-53, ModelMistral.h:23-40."""
    traj = oracle_trajectory(oracle_lib, name, 320, 3, 6)       # 320 tokens: long enough that the step's own appended row (which may round differently) weighs ~1/320
    d = traj.desc
    tol_e2e = 4e-3 if d.layers >= 32 else 3e-3      # the bf16 KV-flip floor grows with depth: measured 1.8-2.1e-3 over 28 layers
    end2end, injected = load_gpus(d)
    for m in (end2end, injected):
        m.forward(traj.prompt)
    e2e, inj = [], []
    for step in range(7):
        lr, le, li = traj.logits[step], end2end.logits(rounded=False), injected.logits(rounded=False)
        e2e.append(rel_err(le, lr)); inj.append(rel_err(li, lr))
        assert e2e[-1] < tol_e2e, (step, e2e)
        if step:
            # the step's OWN appended row cannot be injected (it is computed inside the step): where its fp32 values straddle a bf16 boundary
            # the token attends to a key / value one ulp off, often with a large weight (itself) — measured 2e-5 .. 2.7e-4 per step over
            # 28 layers (test_full_depth_flip_floor_oracle_vs_reordered_oracle measures the same effect with no GPU code involved), against 1.3-1.7e-3 end to end
            assert inj[-1] < 5e-4, (step, inj)
        top2 = np.sort(lr[0])[-2:]
        gap = (top2[1] - top2[0]) / np.abs(lr).max()
        if gap > 2 * tol_e2e:
            assert int(np.argmax(le[0])) == int(np.argmax(lr[0]))
        if gap > 1e-3 and step:
            assert int(np.argmax(li[0])) == int(np.argmax(lr[0]))
        tok = traj.toks[step]
        if step == 6:
            break
        inject_cache(injected, traj, 320 + step)
        if step < 4:
            end2end.sample(GREEDY); injected.sample(GREEDY)
            end2end.forward(tok[None, :]); injected.forward(tok[None, :])
        else:
            force_graph_step(end2end, tok, d.vocab); force_graph_step(injected, tok, d.vocab)
    print(name, "end to end:", ["%.1e" % e for e in e2e], " injected:", ["%.1e" % e for e in inj])
    # the GPU's own cache rows (MFMA prefill + appended rows) against the CPU path's: layer 0 sees identical inputs on both sides -> one
    # bf16 ulp; the last layer's inputs carry 27 / 31 layers of flips (1-2e-3 of the residual stream) -> two ulps plus 1 % of the largest entry
    for layer in (0, d.layers - 1):
        for g_, r_ in zip(end2end.read_kv(0, layer), traj.kv[layer]):
            ulp = 2.0 ** -7
            tol, floor = (ulp, 4e-6) if layer == 0 else (2 * ulp, 1e-2)
            bad = np.abs(g_ - r_) > tol * np.abs(r_) + floor * np.abs(r_).max()
            assert not bad.any(), (layer, int(bad.sum()))


@pytest.mark.parametrize("name,tol", [("llama-3.2-3b", 3e-3), ("mistral-7b-v0.3", 4e-3)])
def test_full_depth_flip_floor_oracle_vs_reordered_oracle(name, tol, oracle_lib):
    """The tolerance of test_full_depth_vs_oracle, justified on the same host in the same run: the CPU oracle against ITSELF with every
    reduction summed last-to-first (tests/test_oracle_reorder.py), bf16 storage, the same geometry, 320-token prompt and 6 steps.  No GPU work — it lives
    in the -m gpu suite because two full-size oracle contexts need the GPU box's host (minutes on the build container's 8 cores).  The two schedules of
    the same code land at 0.8-1.4e-3 (Llama-3.2-3B) / see profiles/r04_kv_flip_floor.txt (Mistral-7B): the floor is above north_star's 1e-3 and the granted
    3e-3 / 4e-3 are ~2x that floor, not slack."""
    from test_oracle_reorder import flip_floor
    flip_floor(name, tol, oracle_lib, 320, 6)


PEAK_PROMPT_SEED = 8      # found with the CPU path alone (tools/verify_checkpoint.py --synthetic-peaked --find-seed): every one of the 257 top-2 gaps >= 9e-3


def test_greedy_tokens_identical_free_running_at_the_bench_operating_point(oracle_lib, tmp_path):
    """north_star: "greedy tokens identical" — literally, at bench.py's operating point: Llama-3.2-1B geometry, 2048-token prompt, then 256 decode steps
    FREE-RUNNING on both sides (== generateSync, GPTEngine.cpp:154-174: prefill, sample, maxNewTokens - 1 steps), no teacher forcing and no tie band.
    With N(0, 0.02^2) weights the top-2 logit gap is under the HIP-vs-CPU distance at one step in twenty, so an id comparison would mostly test the
    tie-break; the *peaked* synthetic checkpoint (tinygpt_amd.synth: four loud lm_head rows and their negations, untied head — the only change against
    the bench model, whose tied head a loud row would turn into a fixed point) makes every step's winner clear.  Held to:
      * at EVERY step the CPU path's top-2 gap exceeds 4x the HIP-vs-CPU logit distance measured at that step (asserted, not assumed).  The distance
        itself is normalised by the largest |logit|, here the largest of 8 loud rows rather than of 128 256 — ~3x smaller than on the bench checkpoint
        for the same hidden-state error — so it reads 2-3e-3 where the bench checkpoint reads 5e-4 (held to 1e-3 by the test above); held to 4e-3;
      * tgx_decode(256) in ONE call (the bench's call: multi-step graph replays) returns the CPU path's 256 ids; a second context stepping one token
        at a time (where the distance is measured at every step) returns them too; so does the C++ engine behind tgx_cli from a checkpoint directory."""
    import copy
    import subprocess
    from host_util import write_model_dir
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd import build, known_desc
    from tinygpt_amd.desc import KNOWN_CONFIGS
    S, STEPS = BENCH_S, 256
    d = copy.deepcopy(known_desc("llama-3.2-1b"))
    d.tied, d.max_ctx, d.max_batch = False, S + STEPS + 8, 1
    prompt = synth.synth_prompt(d.vocab, S, PEAK_PROMPT_SEED)[None, :]
    one_call, stepwise = Model(d, product_backend()), Model(d, product_backend())
    oracle_lib.set_threads(min(32, __import__("os").cpu_count() or 8))
    ref = OracleModel(d)
    for name, bits in synth.synth_checkpoint(d, 1234, 0.02, peaked=True):
        for m in (one_call, stepwise, ref):
            m.upload(name, bits)
    for m in (one_call, stepwise, ref):
        m.finalize()
        m.forward(prompt)
    first = one_call.sample(GREEDY).copy()
    ids_one_call = np.concatenate([first, one_call.decode(STEPS, GREEDY)[:, 0]])          # the bench's call shape
    ids_ref, ids_step, gaps, dist = [], [], [], []
    for step in range(STEPS + 1):
        lr, lg = ref.logits(rounded=False), stepwise.logits(rounded=False)
        top2 = np.partition(lr[0], -2)[-2:]
        gaps.append(float((top2[1] - top2[0]) / np.abs(lr).max()))
        dist.append(rel_err(lg, lr))
        tr, tg = ref.sample(GREEDY), stepwise.sample(GREEDY)
        ids_ref.append(int(tr[0])); ids_step.append(int(tg[0]))
        assert ids_step[-1] == ids_ref[-1], (step, gaps[-1], dist[-1])          # from here on the two contexts would differ
        if step < STEPS:
            ref.forward(tr[None, :]); stepwise.decode(1, GREEDY)
    ref.close(); oracle_lib.set_threads(8)
    print("free-running: min top-2 gap %.2e, max HIP-vs-CPU distance %.2e, %d distinct ids" % (min(gaps), max(dist), len(set(ids_ref))))
    margin = min(g / max(e, 1e-12) for g, e in zip(gaps, dist))
    print("smallest gap / distance ratio of a step: %.1f" % margin)
    assert max(dist) < 4e-3
    assert margin > 4.0, (margin, min(gaps), max(dist))                         # the margin that makes the id comparison meaningful, step by step
    assert len(set(ids_ref)) >= 4                                               # not a fixed point
    np.testing.assert_array_equal(ids_one_call, np.array(ids_ref))              # zero skips
    np.testing.assert_array_equal(np.array(ids_step), np.array(ids_ref))
    one_call.close(); stepwise.close()
    # ---- the same through the C++ engine and its CLI, from a checkpoint directory (3 shards, untied lm_head tensor)
    cfg = dict(KNOWN_CONFIGS["llama-3.2-1b"]); cfg["tie_word_embeddings"] = False
    write_model_dir(str(tmp_path), cfg, 1234, 0.02, shards=3, eos=[128001, 128009], peaked=True)
    _, cli = build.build_host()
    out = subprocess.run([cli, "--model", str(tmp_path), "--device", "mi355x", "--dtype", "bf16", "--max-tokens", str(STEPS + 1), "--temperature", "0", "--top-p", "1",
                          "--prompt-ids", ",".join(str(int(t)) for t in prompt[0])], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("Output ids:")][0]
    np.testing.assert_array_equal(np.array([int(t) for t in line.split(":")[1].split()]), np.array(ids_ref))
