#!/usr/bin/env python3
"""Which kernel serves which prefill regime, and what the regime costs.

    python tools/prefill_regimes.py run  [--models a,b] [--dtypes bf16,fp16] [--layers 2]      # inside rocprofv3 --kernel-trace: prompts S = 5 .. 8192, 60 ms apart
    python tools/prefill_regimes.py map <rocpd .db> <run log>                                   # -> per (model, dtype, S): prefill ms + the kernels that ran

Layers are cut to --layers (the kernels chosen do not depend on the depth); the vocabulary to 4096."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEQS = [5, 24, 48, 100, 130, 200, 300, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192]


def run(argv):
    import argparse
    from tinygpt_amd import known_desc, synth
    from tinygpt_amd.ffi import Model
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="llama-3.2-1b,qwen2.5-0.5b,llama-3.2-3b,mistral-7b-v0.3,qwen3-1.7b,gpt2")
    ap.add_argument("--dtypes", default="bf16,fp16")
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--opts", default="")
    a = ap.parse_args(argv)
    for name in a.models.split(","):
        for dt in a.dtypes.split(","):
            d = copy.deepcopy(known_desc(name, dt))
            d.layers, d.vocab = a.layers, min(d.vocab, 4096)
            seqs = [s for s in SEQS if s + 8 <= (1024 if d.family == "gpt2" else 8200)]
            d.max_ctx = max(seqs) + 8
            m = Model(d).load_synthetic(1234, 0.02).finalize()
            for kv in filter(None, a.opts.split(";")):
                k, v = kv.split("="); m.set_option(k, int(v))
            for S in seqs:
                ids = synth.synth_prompt(d.vocab, S, 1234)[None, :]
                m.synchronize(); time.sleep(0.06)
                m.reset_cache(); m.forward(ids); m.synchronize()            # warm (workspaces, code objects)
                best = 1e9
                for _ in range(3):
                    m.reset_cache(); m.synchronize(); time.sleep(0.06)         # the gap the map splits the trace at
                    t0 = time.perf_counter(); m.forward(ids); m.synchronize(); best = min(best, time.perf_counter() - t0)
                print(f"REGIME {name} {dt} S={S} ms={best * 1e3:.3f}", flush=True)
            m.close()


def kmap(db, log):
    import re, sqlite3
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    groups, cur_g, last_end = [], [], None
    for n, s, e in rows:
        if last_end is not None and s - last_end > 40e6:
            groups.append(cur_g); cur_g = []
        last_end = e
        if "__amd_rocclr" in n: continue                      # the resets' fills and copies
        cur_g.append((n, (e - s) / 1e3))
    groups.append(cur_g)
    groups = [g for g in groups if any("gemm" in n or "gemv" in n or "skinny" in n for n, _ in g)]
    regs = [l.split() for l in open(log) if l.startswith("REGIME")]
    # per regime: 1 warm + 3 timed forwards = the last 3 groups of each regime carry identical kernel sets; take groups in order: the first group also holds load / fill kernels
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void tgx::", "", n))
    gi = len(groups) - 1
    out = []
    for r in reversed(regs):
        g = groups[gi]; gi -= 4 if gi >= 4 else gi
        ks = {}
        for n, us in g:
            k = short(n); ks[k] = (ks.get(k, (0, 0.0))[0] + 1, ks.get(k, (0, 0.0))[1] + us)
        out.append((r, ks))
    for r, ks in reversed(out):
        print(" ".join(r[1:]))
        for k, (c, us) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
            if us > 2.0: print(f"    {c:4d} x {us / c:9.1f} us  {k[:120]}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "map": kmap(sys.argv[2], sys.argv[3])
    else: run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
