"""Full-size oracle runs, computed ONCE per pytest session and shared (the -m gpu suite spent most of its five minutes on the host, repeating
the same full-depth oracle passes in neighbouring tests).

A *trajectory* is what every teacher-forced full-size comparison needs from the CPU path: the fp32 logits after the prompt and after each forced step, the
greedy token the oracle picked at each (the token both sides are then forced with), and the oracle's final K / V cache.  Because every step is forced with
the trajectory's own tokens, the cache rows [0, prompt + t) at step t are a PREFIX of the final cache — `kv_prefix` hands them out for tgx_write_kv.
Keyed by (model, storage dtype, prompt length, prompt seed, steps, reorder, act16, forced-by); the oracle context is closed once the trajectory is stored."""
import copy
import os

import numpy as np

from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY

_CACHE = {}
WIDE = min(32, os.cpu_count() or 8)      # oracle team for full-size passes (tests/conftest.py caps the default at 8: tiny fixtures only pay barriers for more)


class Trajectory:
    def __init__(self, desc, prompt, logits, toks, kv, key=None):
        self.desc, self.prompt, self.logits, self.toks, self.kv, self.key = desc, prompt, logits, toks, kv, key

    def kv_prefix(self, layer, n_rows):
        k, v = self.kv[layer]
        return k[:n_rows], v[:n_rows]


def full_desc(name, max_ctx=384):
    d = copy.deepcopy(known_desc(name))
    d.max_ctx, d.max_batch = max_ctx, 1
    return d


def oracle_trajectory(oracle_lib, name, prompt_len, seed, steps, dtype="bf16", reorder=False, act16=False, forced=None, max_ctx=384, keep_logits=None, kv_layers=None):
    """`forced`: another Trajectory whose tokens drive this one (the reordered schedule is teacher-forced with the forward schedule's tokens);
    `keep_logits(step) -> bool` drops the logits of steps nobody reads (a 272-step run of a 128k vocabulary); `kv_layers`: layers whose cache is kept."""
    from oracle.oracle_ffi import OracleModel
    # everything that shapes what is stored is part of the key: which steps keep their logits (by the filter's name), which layers keep their
    # cache, and the forcing trajectory by ITS key (an id() can be reused after garbage collection)
    key = (name, dtype, prompt_len, seed, steps, reorder, act16, forced.key if forced is not None else None, max_ctx,
           getattr(keep_logits, "__name__", None) if keep_logits is not None else None, tuple(kv_layers) if kv_layers is not None else None)
    if key in _CACHE:
        return _CACHE[key]
    d = full_desc(name, max_ctx)
    d.compute_dtype = dtype
    oracle_lib.set_threads(WIDE)
    try:
        m = OracleModel(d).load_synthetic(1234, 0.02).finalize()
        m.set_reorder(reorder); m.set_act16(act16)
        prompt = synth.synth_prompt(d.vocab, prompt_len, seed)[None, :]
        m.forward(prompt)
        logits, toks = {}, []
        for step in range(steps + 1):
            if keep_logits is None or keep_logits(step):
                logits[step] = m.logits(rounded=False).copy()
            tok = forced.toks[step].copy() if forced is not None else m.sample(GREEDY).copy()
            toks.append(tok)
            if step == steps:
                break
            m.forward(tok[None, :])
        kv = {layer: m.read_kv(0, layer) for layer in (range(d.layers) if kv_layers is None else kv_layers)}
        m.close()
    finally:
        oracle_lib.set_threads(8)
    t = Trajectory(d, prompt, logits, toks, kv, key)
    _CACHE[key] = t
    return t


BENCH_S, BENCH_LAST, BENCH_EVERY = 2048, 272, 32


def bench_checked(step):
    return step < 9 or step % BENCH_EVERY == 0


def bench_range_trajectory(oracle_lib):
    """the oracle over bench.py's workload (rank 0's 2048-token prompt, 272 forced steps): tests/test_hip_parity_bar.py walks the whole range,
    tests/test_hip_fullsize.py's operating-point test reads its first 8 steps"""
    return oracle_trajectory(oracle_lib, "llama-3.2-1b", BENCH_S, 1234, BENCH_LAST, max_ctx=BENCH_S + BENCH_LAST + 8, keep_logits=bench_checked)
