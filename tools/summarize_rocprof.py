#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into one small markdown table per run.

  python tools/summarize_rocprof.py --stats DIR/..._kernel_stats.csv [--pmc FILE.csv ...] --out profiles/NAME.md

PMC notes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x, so the "HBM read" column
is FETCH_SIZE * 1024 * 2 for streaming kernels; WRITE_SIZE is exact.  Counters are per launch
(mean over the launches of the run)."""
import argparse
import collections
import csv
import re


def kname(full):
    m = re.search(r"(k_[a-z_0-9]+)(?:<([^>]*)>)?", full)
    if m:
        return m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")
    return full.split("(")[0][-48:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--title", default="rocprofv3 summary")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    lines = [f"# {a.title}", ""]
    if a.cmd:
        lines += [f"Command: `{a.cmd}`", ""]
    if a.stats:
        lines += ["## Kernel time (rocprofv3 --kernel-trace --stats)", "",
                  "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(a.stats)):
            lines.append(f'| {kname(r["Name"])} | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | '
                         f'{float(r["TotalDurationNs"]) / 1e6:.3f} | {float(r["Percentage"]):.1f} |')
        lines.append("")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in a.pmc:
        for r in csv.DictReader(open(path)):
            agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        counters = sorted({c for d in agg.values() for c in d})
        lines += ["## Counters per launch (rocprofv3 --pmc, separate passes; mean over launches)", "",
                  "| kernel | launches | " + " | ".join(counters) + " | HBM read MB (FETCH*1024*2) | HBM write MB |",
                  "|---|---|" + "---|" * (len(counters) + 2)]
        for k, d in sorted(agg.items()):
            n = max(len(v) for v in d.values())
            vals = [f"{sum(d[c]) / len(d[c]):.4g}" if c in d else "" for c in counters]
            rd = f"{sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024 * 2 / 1e6:.1f}" if "FETCH_SIZE" in d else ""
            wr = f"{sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024 / 1e6:.1f}" if "WRITE_SIZE" in d else ""
            lines.append(f"| {k} | {n} | " + " | ".join(vals) + f" | {rd} | {wr} |")
        lines.append("")
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
