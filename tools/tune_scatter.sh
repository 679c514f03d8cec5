#!/bin/bash
# Runs ON THE GPU BOX: A/B of the k_bin_scatter shapes (PCR_HIP_TUNE_SCATTER).
set -e
OUT=gpurun_out/tune
mkdir -p $OUT
PCR_HIP_TUNE_SCATTER=7 timeout -k 10 600 python -m pytest tests/test_gpu_cabi_parity.py tests/test_gpu_row_bands.py -x -q -m gpu > $OUT/tests_s7.log 2>&1 || (tail -30 $OUT/tests_s7.log; exit 1)
tail -2 $OUT/tests_s7.log
B="python bench.py --no-extras --cpu-sample 0 --steps 10 --warmup 2"
for t in 3 8 6 7; do
  PCR_HIP_TUNE_SCATTER=$t $B --workload C2 > $OUT/c2_s$t.json 2> $OUT/c2_s$t.err
  PCR_HIP_TUNE_SCATTER=$t $B --workload gauss1 --steps 4 > $OUT/g1_s$t.json 2> $OUT/g1_s$t.err
  PCR_HIP_TUNE_SCATTER=$t $B --workload point_avg --grid 16384 --rows 2048 --points 125000000 --steps 4 > $OUT/c5s_s$t.json 2> $OUT/c5s_s$t.err
done
PCR_HIP_TUNE_SCATTER=7 PCR_HIP_TUNE_B=1 $B --workload C2 --steps 2 --warmup 1 > $OUT/c2_prof_s7.json 2> $OUT/c2_prof_s7.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/tune/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["ms_per_step"], d["kernels_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h "cycles per block" $OUT/*.err | sort | uniq -c | sort -rn | head -4
