#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into profiles/<name>.json.

    python tools/summarize_pmc.py <out name> <kernel_stats.csv of the same command> <pmc dir> [<pmc dir> ...]

Per kernel (sntc::* only): launches, mean of every counter per launch, and
hbm_side_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE / WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); both count
the L2's memory-side requests, Infinity-Cache hits included."""
import collections
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path

name, stats, dirs = sys.argv[1], Path(sys.argv[2]), sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sntc" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg_ns, share = {}, {}
for r in csv.DictReader(open(stats)):
    avg_ns[r["Name"]] = float(r["AverageNs"])
    share[r["Name"]] = float(r["Percentage"])
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -share.get(kv[0], 0.0)):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = dict(launches=max(len(x) for x in v.values()), share_of_gpu_time_pct=share.get(k),
             counters_mean_per_launch={c: round(val, 1) for c, val in m.items()})
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_side_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and k in avg_ns:
        # SIMD-cycles available in one launch = avg duration x 2.4 GHz x 256 CUs x 4 SIMDs (stats run of the same command)
        e["avg_launch_ns_from_stats"] = round(avg_ns[k])
        e["mfma_busy_over_simd_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg_ns[k] * 2.4 * 1024), 3)
    if "GRBM_GUI_ACTIVE" in m and k in avg_ns:
        # effective shader clock during the (profiled) launch: GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        e["effective_clock_ghz_profiled"] = round(m["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns[k], 3)
    if m.get("SQ_WAVE_CYCLES"):
        e["wave_cycle_shares"] = {c: round(m[c] / m["SQ_WAVE_CYCLES"], 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in m}
    out[k] = e
root = Path(__file__).resolve().parent.parent
sha = lambda f: hashlib.sha256((root / f).read_bytes()).hexdigest()[:16]
# which kernel source these counters belong to: bench.py marks roofline.traffic stale when the source has changed since
out["_meta"] = dict(gather_gemm_sha16=sha("shallow-ntc_amd/csrc/gather_gemm_kernel.h"), conv_plan_sha16=sha("shallow-ntc_amd/csrc/conv_plan.hip"),
                    bf3_gemm_sha16=sha("shallow-ntc_amd/csrc/bf3_gemm.hip"), rb_fused_sha16=sha("shallow-ntc_amd/csrc/rb_fused.hip"),
                    syn_fused_sha16=sha("shallow-ntc_amd/csrc/syn_fused.hip"))
p = root / "profiles" / f"{name}.json"
p.write_text(json.dumps(out, indent=1))
print(p, len(out) - 1, "kernels")
