#!/usr/bin/env python3
"""verify_checkpoint.py — one command that holds a checkpoint to north_star's bar: the HIP path against the CPU oracle, FREE-RUNNING.

    python tools/verify_checkpoint.py --model-dir D --tokens N [--prompt-len S | --prompt-ids a,b,c] [--dtype bf16]
    python tools/verify_checkpoint.py --synthetic-peaked [--find-seed K] [--prompt-len 2048 --tokens 256]

Both sides load the same tensors (a real HF directory: config.json + model.safetensors[.index.json], ModelLoader.cpp:25-89; or the peaked synthetic
Llama-3.2-1B checkpoint of tinygpt_amd.synth), prefill the same prompt, then generate N tokens greedily, each from its OWN previous token
(== GPTEngine::generateSync, src/engine/GPTEngine.cpp:154-174).  Per step it prints the two ids, the HIP-vs-CPU logit distance (max abs / max abs) and the
CPU path's top-2 gap; it exits 0 iff every id is equal and every distance is under --tol (1e-3).  When the ids part ways it says whether the CPU path's gap at
that step was inside the tie band (4 x the distance) — a tie, not a defect — and stops (the contexts differ from there on).

A checker, like tools/quick_parity.py: it is the only kind of code outside tests/ that touches oracle/ (the product never does, tests/test_abi.py).
--find-seed K (CPU only, no GPU needed): tries prompt seeds 1..K on the peaked synthetic checkpoint and reports those whose every top-2 gap clears --gap
(tests/test_hip_parity_bar.py uses the first good one)."""
import argparse
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir")
    ap.add_argument("--synthetic-peaked", action="store_true", help="Llama-3.2-1B geometry, untied head, tinygpt_amd.synth peaked checkpoint (seed 1234)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=2048)
    ap.add_argument("--prompt-seed", type=int, default=8)
    ap.add_argument("--prompt-ids", default=None, help="comma-separated ids instead of the seeded uniform prompt")
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--gap", type=float, default=4e-3, help="--find-seed: the top-2 gap every step must clear")
    ap.add_argument("--find-seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
    args = ap.parse_args()
    if bool(args.model_dir) == bool(args.synthetic_peaked):
        sys.exit("one of --model-dir / --synthetic-peaked")

    from oracle.oracle_ffi import OracleModel, build_oracle, oracle_backend
    from tinygpt_amd import known_desc, synth
    from tinygpt_amd.ffi import GREEDY
    build_oracle()
    oracle_backend().set_threads(args.threads)
    if args.model_dir:
        from tinygpt_amd.checkpoint import iter_checkpoint
        from tinygpt_amd.desc import load_desc
        d = load_desc(args.model_dir, args.dtype)
        tensors = lambda: iter_checkpoint(args.model_dir)
        strict = False
    else:
        d = copy.deepcopy(known_desc("llama-3.2-1b", args.dtype))
        d.tied = False
        tensors = lambda: synth.synth_checkpoint(d, 1234, 0.02, peaked=True)
        strict = True
    if args.prompt_ids:
        prompt0 = np.array([int(t) for t in args.prompt_ids.split(",")], dtype=np.int64)
    else:
        prompt0 = None
    S = len(prompt0) if prompt0 is not None else args.prompt_len
    d.max_batch = 1
    d.max_ctx = min(d.max_ctx, S + args.tokens + 8) if d.max_ctx else S + args.tokens + 8
    if S + args.tokens > d.max_ctx:
        sys.exit(f"prompt + tokens = {S + args.tokens} exceeds contextSize {d.max_ctx}")

    ref = OracleModel(d)
    for name, bits in tensors():
        ref.upload(name, bits, strict=strict)
    ref.finalize()

    def run_ref(prompt, gpu=None):
        """free-running greedy on the CPU path (and on the GPU in lockstep when given): per step (cpu id, gpu id, distance, gap)"""
        ref.reset_cache(); ref.forward(prompt[None, :])
        if gpu:
            gpu.reset_cache(); gpu.forward(prompt[None, :])
        rows = []
        for step in range(args.tokens):
            lr = ref.logits(rounded=False)
            top2 = np.partition(lr[0], -2)[-2:]
            gap = float((top2[1] - top2[0]) / np.abs(lr).max())
            tr = ref.sample(GREEDY)
            tg, dist = tr, 0.0
            if gpu:
                lg = gpu.logits(rounded=False)
                dist = float(np.abs(lg - lr).max() / np.abs(lr).max())
                tg = gpu.sample(GREEDY)
            rows.append((int(tr[0]), int(tg[0]), dist, gap))
            if int(tr[0]) != int(tg[0]) or step + 1 == args.tokens:
                break
            ref.forward(tr[None, :])
            if gpu:
                gpu.decode(1, GREEDY)
        return rows

    if args.find_seed:
        for seed in range(1, args.find_seed + 1):
            rows = run_ref(synth.synth_prompt(d.vocab, S, seed))
            gaps = np.array([r[3] for r in rows])
            print(f"prompt seed {seed}: min top-2 gap {gaps.min():.2e} (step {int(gaps.argmin())}), {len(set(r[0] for r in rows))} distinct ids -> "
                  f"{'GOOD' if gaps.min() >= args.gap else 'no'}", flush=True)
        return 0

    from tinygpt_amd.ffi import Model, product_backend
    gpu = Model(d, product_backend())                                  # raises if the HIP library is missing
    for name, bits in tensors():
        gpu.upload(name, bits, strict=strict)
    gpu.finalize()
    prompt = prompt0 if prompt0 is not None else synth.synth_prompt(d.vocab, S, args.prompt_seed)
    rows = run_ref(prompt, gpu)
    worst = 0.0
    for step, (tr, tg, dist, gap) in enumerate(rows):
        worst = max(worst, dist)
        print(f"step {step:4d}  cpu {tr:7d}  hip {tg:7d}  distance {dist:.2e}  cpu top-2 gap {gap:.2e}{'' if tr == tg else '   <-- ids differ'}")
    tr, tg, dist, gap = rows[-1]
    ok = tr == tg and worst < args.tol and len(rows) == args.tokens
    if tr != tg:
        print(f"ids part ways at step {len(rows) - 1}: the CPU path's top-2 gap there is {gap:.2e}, the distance {dist:.2e} -> "
              + ("inside the tie band (gap < 4 x distance): a tie, not a defect" if gap < 4 * dist else "OUTSIDE the tie band: a real difference"))
    print(f"{len(rows)} of {args.tokens} tokens compared: ids {'identical' if tr == tg and len(rows) == args.tokens else 'DIFFER'}, max distance {worst:.2e} (bar {args.tol:g}) -> {'OK' if ok else 'FAIL'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
