#!/bin/bash
# Runs ON THE GPU BOX: bash tools/ubench_l2_partial.sh  -- builds tools/ubench_l2_partial_writes.hip, times it, then collects
# WRITE_SIZE per launch (rocprofv3 --pmc in its own run) and prints bytes leaving the L2 per 16-byte record.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/l2pw
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_partial_writes.hip -o $OUT/ubench_l2pw
$OUT/ubench_l2pw | tee $OUT/times.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o t -- $OUT/ubench_l2pw > $OUT/pmc.txt 2> $OUT/pmc.err || echo "pmc failed"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_partial" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
rows.sort()
lines = [l for l in open("$OUT/times.txt") if l.startswith("tiles")]
# three launches per printed line, in order
for i, l in enumerate(lines):
    mine = rows[3 * i: 3 * i + 3]
    if not mine: break
    kib = sum(v for _, _, v in mine) / len(mine)
    print(l.strip(), f"| WRITE_SIZE {kib * 1024 / 1e6:8.1f} MB = {kib * 1024 / 50e6:5.1f} B per record")
PY
