#!/usr/bin/env python3
"""gpurun_out/r03/ (scratch, written on the GPU box by tools/refresh_profiles_r03.sh) -> profiles/r03_* (committed).
Copies the rocprofv3 kernel-stats CSVs and text outputs as they are and condenses the PMC passes of the fused merge
kernel (FETCH_SIZE x1024 x2 on gfx950 as calibrated by tools/traffic_probe.py: profiles/r03_pmc_traffic.txt; WRITE_SIZE
x1024) into one text file with the SQ counters of tools/pmc_ab.sh."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03")
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
        sys.exit("no gpurun_out/r03")
    copy("bench_r03.json", "bench_r03.json")
    copy("prof_bench/**/*kernel_stats.csv", "r03_bench_kernel_stats.csv")
    copy("prof_pipe/**/*kernel_stats.csv", "r03_pipeline_kernel_stats.csv")
    copy("prof_sparse/**/*kernel_stats.csv", "r03_sparse_kernel_stats.csv")
    copy("prof_batch/**/*kernel_stats.csv", "r03_batch_kernel_stats.csv")
    copy("kbench.txt", "r03_kbench.txt")
    copy("kbench_k1.txt", "r03_kbench_k1.txt")
    copy("cli_throughput.txt", "r03_cli_throughput.txt")
    copy("pmc_popstrat.txt", "r03_pmc_popstrat.txt")
    copy("pytest_gpu.txt", "r03_pytest_gpu.txt")
    out = ["# r03: PMC passes of the fused merge + test (kmd_merge_filter) on one 20v20 partition, 104 M records -> 4 M rows",
           "# (rocprofv3 --pmc, one counter set per pass; tools/refresh_profiles_r03.sh, tools/pmc_ab.sh)", ""]
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
        out += ["", "# SQ counters per launch of k_tile_sums<512, 2048, filter, one limb, whole waves, 32-bit sums>, whole GPU (tools/pmc_ab.sh)"]
        out += [l.rstrip() for l in open(f) if l.strip()]
    open(os.path.join(DST, "r03_pmc_tile.txt"), "w").write("\n".join(out) + "\n")
    print("  r03_pmc_tile.txt")
    b = os.path.join(DST, "bench_r03.json")
    if os.path.exists(b):
        j = json.loads(open(b).read().strip().splitlines()[-1])
        p = j["pipeline"]
        print("bench: value %.3e %s, roofline.frac %.3f, pipeline %.3f ms (frac %.3f), overlapped %.3f ms, batched %.3f ms" % (
            j["value"], j["unit"], j["roofline"]["frac"], p["ms"], p["roofline"]["frac"], p["overlapped"]["ms_per_partition"], p["batched"]["ms_per_partition"]))


if __name__ == "__main__":
    main()
