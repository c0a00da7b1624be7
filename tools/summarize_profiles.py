#!/usr/bin/env python3
"""Copy the rocprofv3 evidence of a round from gpurun_out/ (scratch) into profiles/
(tracked): kernel-trace stats, the PMC traffic passes with the gfx950 FETCH_SIZE correction
(MI355X_MICROARCH.md, HBM section) and profiles/traffic.json, which bench.py reports as
roofline.traffic.   usage: python3 tools/summarize_profiles.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(P, exist_ok=True)


def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(G, name, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


if tag == "r01":          # (later rounds copy their bench files with their own collectors: tools/collect_profiles_rNN.py)
    shutil.copy(os.path.join(G, "prof_r01", "bench_kernel_stats.csv"), os.path.join(P, "%s_bench_kernel_stats.csv" % tag))
    for f in ("bench_r01.json", "bench_r01_soa.json", "bench_r01_rows.json", "bench_r01_bh.json"):
        if os.path.exists(os.path.join(G, f)):
            shutil.copy(os.path.join(G, f), os.path.join(P, f.replace("r01", tag)))

fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
lines = ["# %s: PMC passes (rocprofv3 --pmc, one counter set per pass; tools/traffic_probe.py)" % tag, ""]
probe_bytes = 4 << 30
factor = None
for k, v in fetch.items():
    if "k_read_probe" in k:
        m = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        f = probe_bytes / (m * 1024)
        # the filter kernel at 20v20 u32 issues 8-byte non-temporal loads: that probe's factor
        # is the one applied (any other only if it is missing)
        if "2u>, true" in k or factor is None:
            factor = f
        lines.append("read probe %-66s FETCH_SIZE %.1f KB for %d B read -> correction x%.4f" % (k[30:96], m, probe_bytes, f))
for k, v in write.items():
    if "fillBuffer" in k and max(v["WRITE_SIZE"]) > 1e6:
        lines.append("memset 4 GiB: WRITE_SIZE %.1f KB -> correction x%.4f" % (max(v["WRITE_SIZE"]), probe_bytes / (max(v["WRITE_SIZE"]) * 1024)))
out = {}
for k, v in fetch.items():
    if "k_filter" in k:
        fs = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        ws = 0.0
        for k2, v2 in write.items():
            if k2 == k:
                ws = sum(v2["WRITE_SIZE"]) / len(v2["WRITE_SIZE"])
        rd = fs * 1024 * (factor or 2.0)
        wr = ws * 1024
        rows = 39_062_500
        lines += ["", "filter kernel: %s" % k[:100],
                  "  FETCH_SIZE %.1f KB/launch x1024 x%.3f (gfx950 correction, calibrated above) = %.4e B read" % (fs, factor or 2.0, rd),
                  "  WRITE_SIZE %.1f KB/launch x1024 = %.4e B written" % (ws, wr),
                  "  HBM bytes per launch = %.4e  (%.2f B/row over %d rows; algorithmic 168 B/row = %.4e; counts only 160 B/row = %.4e)"
                  % (rd + wr, (rd + wr) / rows, rows, 168.0 * rows, 160.0 * rows)]
        out = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "rows_per_launch": rows,
               "fetch_size_correction": factor or 2.0, "source": "profiles/%s_pmc_traffic.txt" % tag}
for k, v in sq.items():
    if "k_filter" in k:
        n = len(v["SQ_WAVES"])
        lines += ["", "SQ counters per launch (%d launches): " % n + ", ".join("%s=%.4g" % (c, sum(x) / n) for c, x in sorted(v.items()))]
        a = {c: sum(x) / n for c, x in v.items()}
        lines.append("  VALU instructions per wave-row (64 rows): %.0f ; SALU %.0f ; VALU busy / wave cycles = %.1f %% ; waiting on memory = %.1f %%"
                     % (a["SQ_INSTS_VALU"] / (39_062_500 / 64), a["SQ_INSTS_SALU"] / (39_062_500 / 64),
                        100 * a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]))
open(os.path.join(P, "%s_pmc_traffic.txt" % tag), "w").write("\n".join(lines) + "\n")
if out:
    json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
