#!/usr/bin/env python3
"""gpurun_out/r02/ (scratch, written on the GPU box by tools/refresh_profiles_r02.sh) -> profiles/r02_* (committed).
Copies the rocprofv3 kernel-stats CSVs and text outputs as they are and condenses the PMC passes
(FETCH_SIZE / WRITE_SIZE of the fused merge kernel) into one text file, with the unit and gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE in KB, x2 on gfx950 as calibrated
by tools/traffic_probe.py in round 1: profiles/r01_pmc_traffic.txt)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02")
DST = os.path.join(ROOT, "profiles")


def first(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    return hits[0] if hits else None


def copy(pattern, name):
    f = first(pattern)
    if f:
        shutil.copyfile(f, os.path.join(DST, name))
        print("  %-40s <- %s" % (name, os.path.relpath(f, ROOT)))
    else:
        print("  %-40s MISSING (%s)" % (name, pattern))


def pmc_per_kernel(directory, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if want not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    return {k: {c: (v / n[(k, c)], n[(k, c)]) for c, v in d.items()} for k, d in acc.items()}


def main():
    if not os.path.isdir(SRC):
        sys.exit("no gpurun_out/r02")
    os.makedirs(DST, exist_ok=True)
    copy("bench_r02.json", "bench_r02.json")
    copy("prof_bench/**/*kernel_stats.csv", "r02_bench_kernel_stats.csv")
    copy("prof_pipe/**/*kernel_stats.csv", "r02_pipeline_kernel_stats.csv")
    copy("prof_sparse/**/*kernel_stats.csv", "r02_sparse_kernel_stats.csv")
    copy("prof_overlap/**/*kernel_stats.csv", "r02_overlap_kernel_stats.csv")
    copy("overlap.txt", "r02_overlap.txt")
    copy("kbench.txt", "r02_kbench.txt")
    copy("cli_throughput.txt", "r02_cli_throughput.txt")
    copy("pmc_popstrat.txt", "r02_pmc_popstrat.txt")
    copy("pytest_gpu.txt", "r02_pytest_gpu.txt")
    # the fused merge kernel: HBM traffic per launch + SQ counters
    out = ["# r02: PMC passes of the fused merge + test (kmd_merge_filter) on one 20v20 partition, 104 M records -> 4 M rows",
           "# (rocprofv3 --pmc, one counter per pass; tools/refresh_profiles_r02.sh, tools/pmc_tile.sh)", ""]
    traffic = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, d in pmc_per_kernel(os.path.join(SRC, "pmc_" + c), "k_tile_sums").items():
            if c in d:
                kb, launches = d[c]
                corr = 2.0 if c == "FETCH_SIZE" else 1.0
                traffic.setdefault(k, {})[c] = kb * 1024 * corr
                out.append("%s  %s %.1f KB/launch (%d launches) x1024 x%.1f (gfx950) = %.4e B" % (k[:90], c, kb, launches, corr, kb * 1024 * corr))
    for k, t in traffic.items():
        if "FETCH_SIZE" in t and t["FETCH_SIZE"] > 1e8:
            out.append("")
            out.append("merge kernel HBM bytes per launch = %.4e read + %.4e written; algorithmic 12 B x 103 899 205 records = 1.2468e+09 B (x%.3f)"
                       % (t["FETCH_SIZE"], t.get("WRITE_SIZE", 0.0), (t["FETCH_SIZE"] + t.get("WRITE_SIZE", 0.0)) / 1.24679e9))
    f = os.path.join(SRC, "pmc_tile.txt")
    if os.path.exists(f):
        out += ["", "# SQ counters per launch of k_tile_sums (tools/pmc_tile.sh)"] + [l.rstrip() for l in open(f) if "per launch" in l]
    f = os.path.join(SRC, "pmc_tile2.txt")
    if os.path.exists(f):
        vals = {}
        for l in open(f):
            if "per launch" in l:
                name, v = l.split()[0], float(l.split()[1])
                vals[name] = v
        out += ["", "# utilisation of the wide instantiation, whole GPU per launch (tools/pmc_tile2.sh)"] + ["%s %.4g" % kv for kv in sorted(vals.items())]
        cu = vals.get("SQ_BUSY_CU_CYCLES")
        if cu:
            out.append("")
            out.append("per busy CU cycle: LDS active %.0f %% (bank conflicts %.0f %% of that), scalar instructions %.0f %%, vector instructions x 4 cycles / 4 SIMDs %.0f %%"
                       % (100 * vals.get("SQ_LDS_IDX_ACTIVE", 0) / cu, 100 * vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, vals.get("SQ_LDS_IDX_ACTIVE", 1)),
                          100 * vals.get("SQ_INSTS_SALU", 0) / cu, 100 * vals.get("SQ_INSTS_VALU", 0) / cu))
    open(os.path.join(DST, "r02_pmc_tile.txt"), "w").write("\n".join(out) + "\n")
    print("  r02_pmc_tile.txt")
    b = os.path.join(DST, "bench_r02.json")
    if os.path.exists(b):
        j = json.loads(open(b).read().strip().splitlines()[-1])
        print("bench: value %.3e %s, roofline.frac %.3f, pipeline %.3f ms (frac %.3f)" % (
            j["value"], j["unit"], j["roofline"]["frac"], j["pipeline"]["ms"], j["pipeline"]["roofline"]["frac"]))


if __name__ == "__main__":
    main()
