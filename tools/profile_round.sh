#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02'): rocprofv3 kernel-trace stats of the default bench
# workload (C2) and of the other workloads, plus separate PMC passes (FETCH_SIZE / WRITE_SIZE, SQ counters) as
# MI355X_MICROARCH.md prescribes (counters in their own runs, never combined with tracing).
# Outputs CSVs under gpurun_out/prof_<round>/; tools/collect_profiles.py turns them into profiles/<round>_*.md and
# profiles/pmc_traffic.json afterwards (in the build container).
set -e
R=${1:-r02}
WL=${2:-"C4 line16 gauss1 gauss4 gauss16"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
B="python3 bench.py --no-extras --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C2 -o t -- $B --workload C2 --steps 10 --warmup 2 > $OUT/C2_trace.json 2> $OUT/C2_trace.err
echo "trace C2 done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/C2_$c -o t -- $B --workload C2 --steps 3 --warmup 1 > $OUT/C2_$c.json 2> $OUT/C2_$c.err
  echo "pmc $c C2 done"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/C2_sq1 -o t -- $B --workload C2 --steps 3 --warmup 1 > $OUT/C2_sq1.json 2> $OUT/C2_sq1.err || echo "sq1 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/C2_sq2 -o t -- $B --workload C2 --steps 3 --warmup 1 > $OUT/C2_sq2.json 2> $OUT/C2_sq2.err || echo "sq2 failed"
for w in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o t -- $B --workload $w --steps 5 --warmup 1 > $OUT/${w}_trace.json 2> $OUT/${w}_trace.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${w}_$c -o t -- $B --workload $w --steps 2 --warmup 1 > $OUT/${w}_$c.json 2> $OUT/${w}_$c.err
  done
  echo "$w done"
done
find $OUT -name "*.csv" | wc -l
du -sh $OUT
