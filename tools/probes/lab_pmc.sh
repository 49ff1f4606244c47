#!/bin/bash
# effective clock and matrix-pipe duty of every gemm_lab kernel: rocprofv3 --pmc pass (GRBM_GUI_ACTIVE = shader-clock cycles of the dispatch; SQ_VALU_MFMA_BUSY_CYCLES) next to the durations
#   gpurun -- 'bash tools/probes/lab_pmc.sh <out-tag>'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d /tmp/lp -o p -- $R/tools/probes/build/gemm_lab 10 > $O/lab_pmc_run.txt 2>&1
python3 - $(find /tmp/lp -name "*.db" | head -1) > $O/lab_pmc.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name").fetchall()
k = {}
for n, c, cnt, a, d in rows: k.setdefault(n, {"calls": cnt, "us": d / 1e3})[c] = a
print(f"{'kernel':96s} {'calls':>5s} {'us':>8s} {'GHz':>6s} {'mfma duty':>9s}")
for n, v in sorted(k.items(), key=lambda t: -t[1]["us"]):
    g = v.get("GRBM_GUI_ACTIVE", 0); m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    print(f"{n[:96]:96s} {v['calls']:5d} {v['us']:8.1f} {g / v['us'] / 1e3:6.2f} {m / max(g, 1) / 1024:9.3f}   raw mfma {m:.3e} gui {g:.3e}")
PY
cat $O/lab_pmc.txt | cut -c1-200
