#!/usr/bin/env python3
"""serve_sim.py — the kernel contract of continuous batching (per-row lifecycle; paged KV, option kv.budget_tokens) driven the way a
serving loop would drive it, on the real kernels: a seeded stream of requests (prompt and output lengths uniform in given ranges) served on ONE context by

  continuous   every tick: retire the rows that reached their length (tgx_reset_row), admit waiting requests into idle rows while the token budget has room
               (tgx_forward_row + tgx_sample_row), then ONE tgx_decode call of n <= 16 steps for all rows (n = the shortest remaining output)
  static       the reference worker's shape taken to a batch: B requests in, decode until the LONGEST is done (finished rows are retired, their slots stay empty),
               then the next B

and reports generated tokens per second, the mean number of live rows per step and what the cache held.  The reference has neither (its server runs one request at
a time, HttpServer.cpp:118-163; continuous batching and paged attention are README.md:32-34 TODOs): this is the measurement of the kernel half only — no queue, no
HTTP, synthetic weights; greedy unless --sampler.

    python tools/serve_sim.py [--model llama-3.2-1b] [--rows 32] [--requests 200] [--prompt 16,512] [--new 16,256] [--kv-budget 16384] [--policy continuous|static|both]
"""
import argparse, dataclasses, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b")
ap.add_argument("--rows", type=int, default=32)
ap.add_argument("--requests", type=int, default=200)
ap.add_argument("--prompt", default="16,512")
ap.add_argument("--new", default="16,256")
ap.add_argument("--max-ctx", type=int, default=1024)
ap.add_argument("--kv-budget", type=int, default=0, help="paged KV: tokens of cache for all rows together (0 = one max_ctx slab per row)")
ap.add_argument("--policy", default="both", choices=["continuous", "static", "both"])
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--sampler", default="", help="e.g. 'temperature=0.8,top_p=0.9' (default: greedy)")
args = ap.parse_args()
B = args.rows
CFG = GREEDY
if args.sampler:
    from tinygpt_amd.ffi import SamplerCfg
    CFG = SamplerCfg(**{k: (int(v) if k == "top_k" else float(v)) for k, v in (item.split("=") for item in args.sampler.split(","))})
plo, phi = (int(x) for x in args.prompt.split(","))
nlo, nhi = (int(x) for x in args.new.split(","))
assert phi + nhi <= args.max_ctx
desc = dataclasses.replace(known_desc(args.model), max_batch=B, max_ctx=args.max_ctx)
m = Model(desc, product_backend())
if args.kv_budget:
    m.set_option("kv.budget_tokens", args.kv_budget)
m.load_synthetic(1234, 0.02).finalize()
budget = args.kv_budget if args.kv_budget else B * args.max_ctx
blocks = (lambda n: (n + 127) // 128 * 128) if args.kv_budget else (lambda n: args.max_ctx)
rng = np.random.default_rng(args.seed)
reqs = [(int(rng.integers(plo, phi + 1)), int(rng.integers(nlo, nhi + 1))) for _ in range(args.requests)]
prompts = [synth.synth_prompt(desc.vocab, L, 1000 + i) for i, (L, _) in enumerate(reqs)]


def born():
    """a batch of B idle rows (tgx_forward creates the rows, tgx_reset_row retires each)"""
    m.reset_cache()
    m.forward(np.zeros((B, 1), dtype=np.int64)); m.sample(GREEDY)
    for r in range(B):
        m.reset_row(r)


def serve(policy):
    born()
    waiting = list(range(len(reqs)))
    length, target = [0] * B, [0] * B
    produced = steps = live_steps = calls = 0
    peak_tokens = 0
    m.synchronize(); t0 = time.perf_counter()
    while waiting or any(length):
        idle = [r for r in range(B) if not length[r]]
        if policy == "continuous" or len(idle) == B:          # static: a new batch only when the whole previous one is done
            reserved = sum(blocks(t) for t in target if t)
            for r in idle:
                if not waiting:
                    break
                i = waiting[0]
                L, new = reqs[i]
                if reserved + blocks(L + new) > budget:
                    break                                    # the head of the queue waits for room (FIFO)
                waiting.pop(0)
                m.forward_row(r, prompts[i]); m.sample_row(r, CFG, seed=3)
                length[r], target[r] = L, L + new
                reserved += blocks(L + new)
                produced += 1                                 # the first token came from the prefill's logits
        live = [r for r in range(B) if length[r]]
        if not live:
            raise SystemExit("the budget admits no request")
        remaining = [target[r] - length[r] for r in live]
        n = min(16, min(remaining))
        m.decode(n, CFG, seed=3, fetch=False)
        calls += 1; steps += n
        for r in live:
            length[r] += n; produced += n; live_steps += n
        peak_tokens = max(peak_tokens, sum(length))
        for r in live:                                         # a finished row is retired at once under both policies (static: its slot stays empty until the batch is done)
            if length[r] >= target[r]:
                m.reset_row(r); length[r] = target[r] = 0
    m.synchronize(); dt = time.perf_counter() - t0
    print(f"{policy:10s} {len(reqs)} requests, {produced} tokens generated in {dt:.2f} s = {produced / dt:8.0f} tokens/s; {steps} steps in {calls} decode calls, "
          f"{live_steps / max(steps, 1):.1f} live rows per step of {B}; most tokens held at once {peak_tokens} (cache: {budget} tokens"
          f"{', paged' if args.kv_budget else ' as slabs'})", flush=True)


print(f"{desc.name}: {B} rows, max_ctx {args.max_ctx}, prompts {plo}..{phi}, outputs {nlo}..{nhi} tokens, "
      f"{'kv.budget_tokens ' + str(args.kv_budget) if args.kv_budget else 'unpaged'}", flush=True)
for pol in (["continuous", "static"] if args.policy == "both" else [args.policy]):
    serve(pol)
