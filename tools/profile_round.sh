#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh'): rocprofv3 kernel-trace stats of the default bench
# command and of the other workloads, plus separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the default one.
# Outputs CSVs under gpurun_out/prof/; tools/summarize_rocprof.py turns them into profiles/*.md afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
mkdir -p $OUT
python bench.py > $OUT/bench_C2.json 2> $OUT/bench_C2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C2 -o c2 -- python bench.py --cpu-sample 0 > $OUT/C2_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/C2_fetch -o c2 -- python bench.py --cpu-sample 0 --steps 3 --warmup 1 > $OUT/C2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/C2_write -o c2 -- python bench.py --cpu-sample 0 --steps 3 --warmup 1 > $OUT/C2_write.log 2>&1
for w in C4 line16 gauss1 gauss4 gauss16; do
  python bench.py --workload $w --cpu-sample 0 > $OUT/bench_$w.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o t -- python bench.py --workload $w --cpu-sample 0 --steps 5 --warmup 1 > $OUT/${w}_trace.log 2>&1
done
find $OUT -name "*.csv" | head -40
