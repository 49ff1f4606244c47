#!/usr/bin/env python3
"""FETCH_SIZE of the gate_up GEMV (GPT-2: the c_fc GEMV) from a rocprofv3 --pmc FETCH_SIZE rocpd .db -> profiles/pmc_traffic.json.
usage: pmc_traffic_update.py <db> <model> <dtype> <pmc_traffic.json>
FETCH_SIZE is reported in KiB and counts 64 B per 128-byte request on gfx950 (MI355X_MICROARCH.md section HBM): bytes = KiB * 1024 * 2."""
import hashlib, json, os, sqlite3, sys
db, model, dtype, path = sys.argv[1:5]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = 'FETCH_SIZE' group by name").fetchall()
# gemv_kernel<DT, PRO, EPI, ...>: EPI 2 = siluMul (gate_up), EPI 4 = gelu (GPT-2 c_fc)
def epi_of(n):          # gemv_kernel<DT, PRO, EPI, NX, R, PF>
    try:
        return n.split("gemv_kernel<", 1)[1].split(">")[0].split(",")[2].strip()
    except IndexError:
        return ""
pick = [r for r in rows if "gemv_kernel<" in r[0] and epi_of(r[0]) in ("2", "4")]
if not pick:
    sys.exit(f"no gate_up GEMV in {db}: {[r[0][:60] for r in rows][:6]}")
name, calls, kib = max(pick, key=lambda r: r[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("gemv.h", "common.h"):          # == bench.py kernel_source_sha256(): the figure is valid for this kernel source only
    h.update(open(os.path.join(root, "tinygpt_amd", "csrc", "kernels", f), "rb").read())
rec = json.load(open(path))
rec.setdefault("by_config", {})[f"{model}:{dtype}"] = {
    "gateup_bytes_per_launch": int(round(kib * 1024 * 2)), "fetch_size_kib_avg": round(kib, 2), "calls": calls, "kernel": name[:100], "kernel_src_sha256": h.hexdigest(),
    "source": "tools/bench_configs.sh (rocprofv3 --pmc FETCH_SIZE pass of bench.py --no-graph, x2 gfx950 correction), round 6"}
json.dump(rec, open(path, "w"), indent=1)
print(f"{model}:{dtype} gate_up FETCH_SIZE {kib:.1f} KiB x 2 = {kib * 2048 / 1e6:.2f} MB per launch ({calls} launches)")
