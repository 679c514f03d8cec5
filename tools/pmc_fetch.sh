#!/bin/bash
# Runs ON THE GPU BOX: HBM fetch / write bytes per kernel for one workload (two --pmc passes, no tracing): bash tools/pmc_fetch.sh gauss16
set -e
W=${1:-gauss16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pf_$W
mkdir -p $OUT
B="python3 bench.py --no-extras --cpu-sample 0 --workload $W --steps 2 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/a -o t -- $B > $OUT/a.json 2> $OUT/a.err || echo "set a failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/b -o t -- $B > $OUT/b.json 2> $OUT/b.err || echo "set b failed"
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(dict)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    tmp = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m: tmp[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in tmp.items(): agg[k][c] = sum(v) / len(v)
for k in sorted(agg):
    # FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction x2 on FETCH (MI355X_MICROARCH.md)
    f = agg[k].get("FETCH_SIZE", 0) * 1024 * 2 / 1e6; w = agg[k].get("WRITE_SIZE", 0) * 1024 / 1e6
    print(f"{k:24s} fetch {f:9.1f} MB   write {w:9.1f} MB")
PY
