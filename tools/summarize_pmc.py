#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into profiles/<tag>_pmc_summary.json.

    python tools/summarize_pmc.py r01 gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write

Per kernel (sntc::* only): launches, mean of every counter per launch, and
hbm_side_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE / WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); both count
the L2's memory-side requests, Infinity-Cache hits included."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

tag, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sntc" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg_ns = {}
stats = Path(__file__).resolve().parent.parent / "profiles" / f"{tag}_decode_only_kernel_stats.csv"
if len(sys.argv) > 2 and sys.argv[-1].endswith(".csv"):      # explicit kernel-stats file as the last argument
    stats, dirs = Path(sys.argv[-1]), dirs[:-1]
if stats.exists():
    for r in csv.DictReader(open(stats)):
        avg_ns[r["Name"]] = float(r["AverageNs"])
out = {}
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = dict(launches=max(len(x) for x in v.values()), counters_mean_per_launch={c: round(val, 1) for c, val in m.items()})
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
import hashlib
root = Path(__file__).resolve().parent.parent
# which kernel source these counters belong to: bench.py marks roofline.traffic stale when the source has changed since
out["_meta"] = dict(gather_gemm_sha16=hashlib.sha256((root / "shallow-ntc_amd/csrc/gather_gemm.hip").read_bytes()).hexdigest()[:16],
                    conv_plan_sha16=hashlib.sha256((root / "shallow-ntc_amd/csrc/conv_plan.hip").read_bytes()).hexdigest()[:16])
p = Path(__file__).resolve().parent.parent / "profiles" / f"{tag}_pmc_summary.json"
p.write_text(json.dumps(out, indent=1))
print(p, len(out), "kernels")
