"""Build-time properties of the hand-written kernels that no numerics test sees (no GPU needed: hipcc cross-compiles gfx950 and reports its own resource usage)."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_attention_kernels_do_not_spill():
    """A spilling kernel is correct and slow.  Round 4's rewrite of the decode attention loop (one softmax update per block of four wave-loads) left the 16-wave
    multi-head forms of a batched step with 38-100 spilled registers — B = 12 / 16 went from 0.91 / 0.97 to 1.03 / 1.11 ms per step under a green suite; they now
    run two / one wave-load per block.  Known and accepted: 4 registers in the four-heads-per-workgroup form that also finishes the QKV product."""
    import check_spills
    rows = check_spills.spills("attn")
    assert len(rows) > 50
    bad = [r for r in rows if r["spill"] > 8 or r["scratch"] > 32]
    assert not bad, bad


def test_skinny_kernels_in_use_have_no_scratch():
    """The panel kernel that stages fp32 rows with the RMSNorm and splits them into three 16-bit terms (the bf16 QKV product of 17-32-row prompts) sits at the edge of
    hipcc's unrolling budget: two added instructions in its staging lambda (round 4, a first form of act.round16) sent its register arrays to scratch memory — prompts of
    20-32 tokens 1.2 -> 1.7 ms, again under a green suite.  Known users of scratch, none of them launched: the fp16 three-term forms (fp16 takes two terms) and the
    four-block 256-k panel (skinny.hip routes it to 128-k panels)."""
    import check_spills
    rows = check_spills.spills("skinny")
    bad = [r for r in rows if (r["spill"] > 8 or r["scratch"] > 32)
           and not ("skinny_gemm_kernel<1, " in r["name"] and ", 3, 0, 2>" in r["name"]) and not (", 4, 3, 0, 0>" in r["name"])]
    assert not bad, bad
