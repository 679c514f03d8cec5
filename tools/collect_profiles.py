#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed evidence:
profiles/<round>_<workload>_rocprof.md (kernel times + HBM bytes per launch) and profiles/pmc_traffic.json (what
bench.py quotes as roofline.traffic -- keyed by the fingerprint of the kernel sources it was measured on).

    python tools/collect_profiles.py prof_r02 r02
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kname(full):
    m = re.search(r"(k_[a-z_0-9]+)", full)
    return m.group(1) if m else full.split("(")[0][-40:]


def pmc_means(path):
    """{kernel base name: {counter: [per-launch values]}}; template variants of one kernel (e.g. a full-chunk and a
    ragged-end launch) are kept apart and the variant with the larger mean represents the name."""
    per_variant = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per_variant[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg = {}
    for full, d in per_variant.items():
        base = kname(full)
        size = max(sum(v) / len(v) for v in d.values())
        if base not in agg or size > agg[base][0]:
            agg[base] = (size, d)
    return {k: v[1] for k, v in agg.items()}


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    points = {"C2": 50_000_000}
    traffic = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), corrected as "
                        "MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE KiB * 1024 * 2 for wide streaming reads; "
                        "WRITE_SIZE KiB * 1024).  bench.py quotes a figure only while csrc_sha matches the kernel sources.",
               "csrc_sha": None}
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    for wl_dir in sorted(glob.glob(os.path.join(src, "*_FETCH_SIZE"))):
        wl = os.path.basename(wl_dir)[:-len("_FETCH_SIZE")]
        stats = glob.glob(os.path.join(src, wl, "**", "*kernel_stats.csv"), recursive=True)
        pmcs = glob.glob(os.path.join(src, wl + "_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True) + \
            glob.glob(os.path.join(src, wl + "_WRITE_SIZE", "**", "*counter_collection.csv"), recursive=True)
        out = os.path.join(ROOT, "profiles", f"{rnd}_{wl}_rocprof.md")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), "--out", out,
               "--title", f"Round {rnd[1:]}, {wl}: 50 M points, 4096^2 (bench.py --workload {wl} --no-extras)",
               "--cmd", f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --cpu-sample 0 --workload {wl}  "
                        "(PMC: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)"]
        if stats:
            cmd += ["--stats", stats[0]]
        if pmcs:
            cmd += ["--pmc"] + pmcs
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        fetch = pmc_means(os.path.join(src, wl + "_FETCH_SIZE"))
        write = pmc_means(os.path.join(src, wl + "_WRITE_SIZE"))
        entry = {"points_per_launch": 50_000_000}
        for k in sorted(set(fetch) | set(write)):
            if not k.startswith("k_"):
                continue
            rd = fetch.get(k, {}).get("FETCH_SIZE", [0.0])
            wr = write.get(k, {}).get("WRITE_SIZE", [0.0])
            entry[k] = {"read_bytes": int(sum(rd) / len(rd) * 1024 * 2), "write_bytes": int(sum(wr) / len(wr) * 1024)}
        traffic[wl] = entry
        print(wl, {k: v for k, v in entry.items() if k != "points_per_launch"})
    # fingerprint of the kernel sources (same function as bench.py)
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pointcloud-raster_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    traffic["csrc_sha"] = h.hexdigest()[:16]
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    # SQ counters, if collected
    sq = collections.defaultdict(dict)
    for sub in ("C2_sq1", "C2_sq2"):
        for k, d in pmc_means(os.path.join(src, sub)).items():
            for c, v in d.items():
                sq[k][c] = sum(v) / len(v)
    if sq:
        cols = sorted({c for d in sq.values() for c in d})
        lines = [f"# Round {rnd[1:]}, C2: SQ counters per launch (rocprofv3 --pmc, two separate passes, mean over launches)", "",
                 "| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |", "|---|" + "---|" * len(cols)]
        for k in sorted(sq):
            if k.startswith("k_"):
                lines.append(f"| {k} | " + " | ".join(f"{sq[k].get(c, float('nan')):.3g}" for c in cols) + " |")
        open(os.path.join(ROOT, "profiles", f"{rnd}_c2_sq_counters.md"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()
