#!/usr/bin/env python3
"""Register spills of the hand-written kernels, from hipcc's own resource remarks (no GPU needed).

    python tools/check_spills.py [attn decode sampler ...]        # translation units of tinygpt_amd/csrc; default: attn

A kernel that spills still computes the right thing — which is how round 4's rewrite of the decode attention loop left the 16-wave multi-head forms of a
batched step with 38-100 spilled registers (B = 12 / 16: 0.91 / 0.97 -> 1.03 / 1.11 ms per step) under a green test suite.  tests/test_build.py runs this for attn.hip."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import build as B


def spills(tu):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [B.HIPCC] + [f for f in B.FLAGS if f != "-shared"] + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, tu + ".hip"), "-o", os.path.join(tmp, "x.o")]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1), "vgpr": 0, "spill": 0, "scratch": 0}
            out.append(cur)
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


if __name__ == "__main__":
    for tu in (sys.argv[1:] or ["attn"]):
        rows = spills(tu)
        bad = [r for r in rows if r["spill"] or r["scratch"]]
        print(f"{tu}.hip: {len(rows)} kernels, {len(bad)} with spills / scratch")
        for r in bad:
            print(f"  spill {r['spill']:4d}  scratch {r['scratch']:4d} B/lane  vgpr {r['vgpr']:3d}  {r['name'][:150]}")
