#!/usr/bin/env python3
"""gpurun_out/r06/ (scratch, written on the GPU box by tools/refresh_profiles_r06.sh) -> profiles/r06_* (committed).
Copies the rocprofv3 kernel-stats CSVs and the text outputs as they are, condenses the PMC passes, and writes
profiles/traffic.json: K1's HBM bytes per launch (tools/summarize_profiles.py, FETCH_SIZE calibrated on read probes in the
same passes) together with the kernel's average duration in the committed kernel-trace CSV and the HIP-event average the
profiled run itself reported -- the two must agree to 3 % (checked here), and bench.py quotes them as roofline.profiled."""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
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
    return f


def json_line(path):
    if not path or not os.path.exists(path):
        return None
    for l in open(path):
        if l.startswith("{"):
            return json.loads(l)
    return None


def main():
    if not os.path.isdir(SRC):
        sys.exit("no gpurun_out/r06")
    copy("bench_r06.json", "bench_r06.json")
    copy("prof_bench/**/*kernel_stats.csv", "r06_bench_kernel_stats.csv")
    copy("c3/trace/**/*kernel_stats.csv", "r06_c3_kernel_stats.csv")
    copy("prof_batch/**/*kernel_stats.csv", "r06_c3_batch_kernel_stats.csv")
    copy("prof_sparse/**/*kernel_stats.csv", "r06_sparse_kernel_stats.csv")
    copy("prof_mixed/**/*kernel_stats.csv", "r06_mixed_kernel_stats.csv")
    f = os.path.join(SRC, "pmc_tile_sparse.txt")
    if os.path.exists(f):
        lines = ["# r06: SQ counters per launch of the merge kernel (tools/pmc_ab.sh), whole GPU: rows of 2.9 records (36 M rows, 104 M records, the",
                 "# 1024-thread / 4096-slot shape), rows of 7.8 records (13 M rows, 104 M records, 512 / 2048), and one MIXED configs[2]-size partition", ""]
        lines += [l.rstrip() for l in open(f) if l.strip() and "simple_timer" not in l]
        open(os.path.join(DST, "r06_pmc_tile_sparse.txt"), "w").write("\n".join(lines) + "\n")
        print("  r06_pmc_tile_sparse.txt")
    f = os.path.join(SRC, "hunt.txt")
    if os.path.exists(f):
        lines = ["# r06: tools/hunt_abort.sh 6 stress6 stress -- the concurrent single-call path in fresh processes (6 host threads, fresh streams,",
                 "# kmd_release_cache between rounds: the configuration that died 1 run in 3 before the fix; then 3 threads on streams they keep)", ""]
        lines += [l.rstrip() for l in open(f) if l.strip()]
        open(os.path.join(DST, "r06_hunt.txt"), "w").write("\n".join(lines) + "\n")
        print("  r06_hunt.txt")
    f = os.path.join(SRC, "ab_k2t.txt")
    if os.path.exists(f):
        lines = ["# r06: the fused merge (kmd_merge_filter) before / after round 6's second half of the merge kernel, same lease (tools/r06_ab.sh):",
                 "# r6_base2 = round 5's kmd_tilemerge.hip linked with this round's other objects; libkmdiff_hip = the shipped library.",
                 "# whole call (best of the iterations) | kernels of the call, average duration under rocprofv3 --kernel-trace --stats", ""]
        lines += [l.rstrip() for l in open(f) if l.strip() and "simple_timer" not in l and "amdgpu.ids" not in l]
        dst = os.path.join(DST, "r06_ab_k2t.txt")
        if os.path.exists(dst):                      # the hand-kept blocks of experiments that were not shipped ("# (n) ...") stay
            kept = open(dst).read().split("\n")
            first = next((i for i, l in enumerate(kept) if l.startswith("# (")), None)
            if first is not None: lines += [""] + [l for l in kept[first:]]
        open(dst, "w").write("\n".join(lines).rstrip("\n") + "\n")
        print("  r06_ab_k2t.txt")
    copy("kbench.txt", "r06_kbench.txt")
    copy("kbench_k1.txt", "r06_kbench_k1.txt")
    copy("pmc_popstrat.txt", "r06_pmc_popstrat.txt")
    copy("pytest_gpu.txt", "r06_pytest_gpu.txt")
    out = []
    notes = {"cli_throughput_packer.txt": " (LD_PRELOAD of round 5's library: kmd_pack_records is this round's, the block packer behind it round 5's)",
             "cli_throughput_release.txt": " (tools/archive/r06_cli_run.sh, an earlier lease of the round, before the page-locked arrays were sized to 5/8 of the file: the staging arrays released in the background against KMD_SYNC_RELEASE=1, alternating)"}
    for name in ("cli_throughput.txt", "cli_throughput_ab.txt", "cli_throughput_packer.txt", "cli_throughput_release.txt"):
        f = os.path.join(SRC, name)
        if os.path.exists(f):
            out += ["# " + name + notes.get(name, ""), ""] + [l.rstrip() for l in open(f) if "amdgpu.ids" not in l] + [""]
    if out:
        open(os.path.join(DST, "r06_cli_throughput.txt"), "w").write("\n".join(out))
        print("  r06_cli_throughput.txt")
    # the merge kernel on a whole configs[2] partition: traffic + SQ counters
    out = ["# r06: kmd_merge_filter on ONE WHOLE configs[2] partition (39 062 500 rows, 1 014 558 591 records = 12.175 GB of streams)",
           "# (tools/prof_c3.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in passes of their own; tools/pmc_ab.sh: SQ counters)", ""]
    f = os.path.join(SRC, "c3_summary.txt")
    if os.path.exists(f):
        lines = [l.rstrip() for l in open(f) if l.strip() and "simple_timer" not in l]
        out += lines
        rd = wr = None
        for l in lines:
            if l.startswith("FETCH_SIZE") and "k_tile_sums<1024" in l:
                rd = float(l.split("-> ")[1].split(" B")[0])
            if l.startswith("WRITE_SIZE") and "k_tile_sums<1024" in l:
                wr = float(l.split("-> ")[1].split(" B")[0])
        if wr is None:                                   # (not among the pass's top four: straight from its CSV)
            acc, n = 0.0, 0
            for c in glob.glob(os.path.join(SRC, "c3", "pmc_WRITE_SIZE", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(c)):
                    if "k_tile_sums<1024" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
                        acc += float(r["Counter_Value"]); n += 1
            if n:
                wr = acc / n * 1024
                out.append("WRITE_SIZE k_tile_sums<1024, 4096u, true, false, true, true> %.1f KB/launch (%d launches) -> %.4e B (x1024 x1)" % (acc / n, n, wr))
        if rd:
            alg = 12.0 * 1014558591
            out += ["", "merge kernel HBM bytes per launch = %.4e read + %.4e written; algorithmic 12 B x 1 014 558 591 records = %.4e B -> x%.3f"
                    % (rd, wr or 0.0, alg, (rd + (wr or 0.0)) / alg)]
    f = os.path.join(SRC, "pmc_tile_sq.txt")
    if os.path.exists(f):
        out += ["", "# SQ counters per launch of k_tile_sums<1024, 4096, filter, one limb, whole waves, 32-bit sums>, whole GPU (tools/pmc_ab.sh)"]
        out += [l.rstrip() for l in open(f) if l.strip()]
    open(os.path.join(DST, "r06_pmc_tile.txt"), "w").write("\n".join(out) + "\n")
    print("  r06_pmc_tile.txt")
    # K1 traffic (gpurun_out/pmc_fetch, pmc_write) -> r06_pmc_traffic.txt + traffic.json
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "r06"], stdout=subprocess.DEVNULL)
    tpath = os.path.join(DST, "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    csv_path = os.path.join(DST, "r06_bench_kernel_stats.csv")
    prof = json_line(os.path.join(SRC, "prof_bench.json"))
    if os.path.exists(csv_path) and prof:
        rows = [r for r in csv.DictReader(open(csv_path)) if "k_filter_soa" in r["Name"]]
        us = float(rows[0]["AverageNs"]) / 1e3
        ev_ms = prof["roofline"]["avg_kernel_ms"]
        dev = abs(us * 1e-3 - ev_ms) / ev_ms
        print("  k_filter_soa: CSV average %.1f us over %s calls, HIP events of the same (profiled) run %.1f us: %.2f %% apart" % (us, rows[0]["Calls"], ev_ms * 1e3, 100 * dev))
        assert dev <= 0.03, "kernel-trace CSV and HIP events of the profiled run differ by more than 3 %"
        tj.update({"profiled_kernel_us": us, "profiled_events_ms": ev_ms, "profiled_calls": int(rows[0]["Calls"]), "layout": "tiled",
                   "profiled_source": "profiles/r06_bench_kernel_stats.csv (k_filter_soa<unsigned int, 2>, rocprofv3 --kernel-trace --stats of "
                                      "`bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline`)"})
        json.dump(tj, open(tpath, "w"), indent=1)
    b = json_line(os.path.join(DST, "bench_r06.json"))
    if b:
        p = b["pipeline"]
        print("bench: value %.3e %s, roofline.frac %.3f (events), pipeline %.3f ms (frac %.3f), overlapped %.3f ms (%.3f), batched %.3f ms (%.3f), small %.3f ms" % (
            b["value"], b["unit"], b["roofline"]["frac"], p["ms"], p["roofline"]["frac"], p["overlapped"]["ms_per_partition"], p["overlapped"]["roofline"]["frac"],
            p["batched"]["ms_per_partition"], p["batched"]["roofline"]["frac"], p["small"]["ms"]))
        if "sparse" in p and "feed_inclusive" in p:
            print("       sparse (mixed) %.3f ms (frac %.3f), batched %.3f; feed-inclusive %.1f ms per partition = %.3e k-mers/s, %.2f of the link's ceiling (%.1f GB/s)" % (
                p["sparse"]["ms"], p["sparse"]["roofline"]["frac"], p["sparse"]["batched"]["roofline"]["frac"], p["feed_inclusive"]["ms_per_partition"],
                p["feed_inclusive"]["kmers_per_s"], p["feed_inclusive"]["frac_of_link_ceiling"], p["feed_inclusive"]["link_GBs"]))


if __name__ == "__main__":
    main()
