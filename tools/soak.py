#!/usr/bin/env python3
"""Determinism soak: the same work repeated many times must give bit-identical results (a race in an LDS-DMA ring, a counted wait one short or a
missing barrier shows up as a rare mismatch long before it shows up in a parity test).  Prefill of 2048 / 3200 / 300 tokens x N, batched decode x 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model
import copy
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = copy.deepcopy(known_desc(name)); d.max_ctx = 4096; d.max_batch = 128
m = Model(d)
if os.environ.get("TGX_KV_BUDGET"):          # the same soak on a paged KV cache (option kv.budget_tokens)
    m.set_option("kv.budget_tokens", int(os.environ["TGX_KV_BUDGET"]))
m.load_synthetic(1234, 0.02).finalize()
bad = 0
for S in (2048, 3200, 1024, 1280, 700, 300, 40, 24, 12):      # (1024 / 1280 / 700: round 4's K slabs on the eight-wave kernel and 128 x 128 gate_up tiles)
    ids = synth.synth_prompt(d.vocab, S, 7)[None, :]
    ref = None
    t0 = time.time()
    for r in range(reps if S >= 300 else 3 * reps):
        m.reset_cache(); m.forward(ids)
        lg = m.logits(False).copy()
        if ref is None: ref = lg
        elif not np.array_equal(ref, lg): bad += 1; print(f"MISMATCH prefill S={S} rep {r}: max diff {np.abs(ref - lg).max():.3e}", flush=True)
    print(f"prefill S={S}: {reps if S >= 300 else 3 * reps} repetitions, {time.time() - t0:.1f} s", flush=True)
for B in (3, 8, 16):
    ids = np.stack([synth.synth_prompt(d.vocab, 64, 11 + b) for b in range(B)])
    outs = []
    for r in range(3):
        m.reset_cache(); m.forward(ids); m.sample(GREEDY)
        outs.append(m.decode(300, GREEDY).copy())
    for r in (1, 2):
        if not np.array_equal(outs[0], outs[r]): bad += 1; print(f"MISMATCH decode B={B} run {r}", flush=True)
    print(f"decode B={B}: 3 x 300 steps equal: {all(np.array_equal(outs[0], o) for o in outs)}", flush=True)
# round 4: the K-sliced o_proj (fixed-point atomics into accumulators that must return to zero every layer) under repetition, across the attention forms
ids = synth.synth_prompt(d.vocab, 700, 5)[None, :]
outs = []
t0 = time.time()
for r in range(3):
    m.reset_cache(); m.forward(ids); m.sample(GREEDY)
    outs.append(m.decode(1500, GREEDY).copy())       # contexts 701 .. 2200: direct form, then split form with the sliced o_proj
    m.synchronize()
for r in (1, 2):
    if not np.array_equal(outs[0], outs[r]): bad += 1; print(f"MISMATCH sliced o_proj run {r}", flush=True)
print(f"batch 1, 3 x 1500 steps equal: {all(np.array_equal(outs[0], o) for o in outs)}  ({time.time() - t0:.1f} s)", flush=True)
for B in (12, 16):
    ids = np.stack([synth.synth_prompt(d.vocab, 1500, 21 + b) for b in range(B)])     # beyond the batch-1 direct limit: the batch form with 2 heads per workgroup
    outs = []
    for r in range(3):
        m.reset_cache(); m.forward(ids); m.sample(GREEDY)
        outs.append(m.decode(200, GREEDY).copy())
    for r in (1, 2):
        if not np.array_equal(outs[0], outs[r]): bad += 1; print(f"MISMATCH long-context decode B={B} run {r}", flush=True)
    print(f"decode B={B} at context 1500+: 3 x 200 steps equal: {all(np.array_equal(outs[0], o) for o in outs)}", flush=True)
# round 3, later: batches of 24-64 rows — the matrix-core attention with the QKV finish in its prologue (the workgroup's clamped K / V loads race with its own
# store of the new row and must never be used), four-block skinny products, 64-row passes
for B, plen in ((24, 600), (32, 600), (48, 300), (64, 600), (64, 70), (100, 200), (128, 500)):
    ids = np.stack([synth.synth_prompt(d.vocab, plen, 31 + b) for b in range(B)])
    outs = []
    t0 = time.time()
    for r in range(3):
        m.reset_cache(); m.forward(ids); m.sample(GREEDY)
        outs.append(m.decode(250, GREEDY).copy())
    for r in (1, 2):
        if not np.array_equal(outs[0], outs[r]): bad += 1; print(f"MISMATCH decode B={B} prompt {plen} run {r}", flush=True)
    print(f"decode B={B} from context {plen}: 3 x 250 steps equal: {all(np.array_equal(outs[0], o) for o in outs)}  ({time.time() - t0:.1f} s)", flush=True)
for S in (33, 48, 64, 100, 128):
    ids = synth.synth_prompt(d.vocab, S, 3)[None, :]
    ref = None
    for r in range(3 * reps):
        m.reset_cache(); m.forward(ids)
        lg = m.logits(False).copy()
        if ref is None: ref = lg
        elif not np.array_equal(ref, lg): bad += 1; print(f"MISMATCH prefill S={S} rep {r}", flush=True)
    print(f"prefill S={S}: {3 * reps} repetitions", flush=True)
# round 4: batch-1 steps — fixed-point atomic sums (attention + o_proj in one launch up to 640 keys, K-sliced o_proj beyond): a generation that crosses both form limits, 3 x
for opt in (0, 1):
    m.set_option("act.round16", opt)
    ids = synth.synth_prompt(d.vocab, 200, 5)[None, :]
    outs = []
    for r in range(3):
        m.reset_cache(); m.forward(ids); m.sample(GREEDY)
        outs.append((m.decode(900, GREEDY).copy(), m.logits(False).copy()))
    ok = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs)
    if not ok: bad += 1
    print(f"decode B=1 act.round16={opt}: 3 x 900 steps from context 200 (ids and final logits) equal: {ok}", flush=True)
m.set_option("act.round16", 0)
# round 6: sampled steps — the draws of a batch's rows in one launch, the step counter moved by the row that completes the batch's count, the tail that draws
from tinygpt_amd.ffi import SamplerCfg
for B, cfg in ((1, SamplerCfg(0.8, 0, 0.9, 0.0)), (2, SamplerCfg(0.8, 50, 0.9, 0.0)), (8, SamplerCfg(0.8, 0, 0.9, 0.0)), (8, SamplerCfg(1.0, 0, 1.0, 0.05)), (32, SamplerCfg(0.8, 50, 0.9, 0.05)), (32, SamplerCfg(0.7, 0, 0.9, 0.0))):
    ids = np.stack([synth.synth_prompt(d.vocab, 200, 41 + b) for b in range(B)])
    outs = []
    for r in range(3):
        m.reset_cache(); m.forward(ids); m.sample(cfg, seed=5)
        outs.append(m.decode(300, cfg, seed=5).copy())
    ok = all(np.array_equal(outs[0], o) for o in outs)
    if not ok: bad += 1
    print(f"sampled decode B={B} (T {cfg.temperature} top-k {cfg.top_k} top-p {cfg.top_p} min-p {cfg.min_p}): 3 x 300 steps equal: {ok}; distinct ids {len(np.unique(outs[0]))}", flush=True)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
