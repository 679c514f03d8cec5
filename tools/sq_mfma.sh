#!/bin/bash
# Runs ON THE GPU BOX: MFMA-pipe and clock counters for one workload (separate --pmc passes, no tracing): bash tools/sq_mfma.sh gauss16
set -e
W=${1:-gauss16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sqm_$W
mkdir -p $OUT
B="python3 bench.py --no-extras --cpu-sample 0 --workload $W --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o t -- $B > $OUT/a.json 2> $OUT/a.err || echo "set a failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/b -o t -- $B > $OUT/b.json 2> $OUT/b.err || echo "set b failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k -o t -- $B > $OUT/k.json 2> $OUT/k.err || echo "trace failed"
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
    print(k, {c.replace("SQ_", ""): f"{v:.4g}" for c, v in sorted(agg[k].items())})
for f in glob.glob("$OUT/k/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
