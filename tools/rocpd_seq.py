#!/usr/bin/env python3
"""Per-position kernel durations of the decode step from a rocprofv3 rocpd .db: the last N dispatches whose names look like the
decode layer sequence (qkv, attn, combine, o_proj, gate_up, down) are grouped by position in the 6-launch layer pattern."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
def cls(n):
    if "attn_decode_kernel" in n: return "attn"
    if "attn_combine" in n: return "combine"
    if "gemv_kernel<0, 1, 0" in n: return "qkv"
    if "gemv_kernel<0, 1, 2" in n: return "gate_up"
    if "gemv_kernel<0, 1, 3" in n: return "lm_head"
    if "gemv_kernel<0, 0, 1" in n: return "resid"
    if "finalize" in n: return "finalize"
    return "other"
seq = [(cls(n), s, e) for n, s, e in rows]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i in range(1, len(seq) - 1):
    c, s, e = seq[i]
    if c == "resid":
        c = "o_proj" if seq[i - 1][0] == "combine" else ("down" if seq[i - 1][0] == "gate_up" else "resid?")
    if c in ("other", "finalize"): continue
    dur[c].append((e - s) / 1e3); gap[c].append((seq[i + 1][1] - e) / 1e3)
print(f"{'class':10s} {'n':>6s} {'dur avg us':>11s} {'gap after us':>13s}")
tot = 0
for c in ("qkv", "attn", "combine", "o_proj", "gate_up", "down", "lm_head"):
    if dur[c]:
        d = sorted(dur[c]); g = sorted(gap[c])
        print(f"{c:10s} {len(d):6d} {sum(d) / len(d):11.2f} {g[len(g) // 2]:13.2f}")
