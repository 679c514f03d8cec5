#!/bin/bash
# Runs ON THE GPU BOX: SQ counters for one workload (separate --pmc passes, no tracing):  bash tools/sq_counters.sh line16
set -e
W=${1:-line16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$W
mkdir -p $OUT
B="python3 bench.py --no-extras --cpu-sample 0 --workload $W --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -o t -- $B > $OUT/a.json 2> $OUT/a.err || echo "set a failed"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/b -o t -- $B > $OUT/b.json 2> $OUT/b.err || echo "set b failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/c -o t -- $B > $OUT/c.json 2> $OUT/c.err || echo "set c failed"
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
    print(k, {c.replace("SQ_", ""): f"{v:.3g}" for c, v in sorted(agg[k].items())})
PY
