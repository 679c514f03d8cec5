#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/tune_scatter.sh'): A/B of the Point binning on C2, a C5 shard and the Gaussian's
# index records -- production shape (1024 x 28 points through a 64 KB window) vs round 1's shape (PCR_HIP_TUNE_SCATTER=3)
# vs the opt-in one-pass sort (PCR_HIP_ONE_PASS=1), plus k_bin_scatter's phase profile (PCR_HIP_TUNE_B=1).
# The shapes that were measured and pruned (two workgroups per CU, 20 / 24 / 32 points per thread, 64-byte aligned runs)
# are recorded in profiles/r02_tune_scatter.md.
set -e
OUT=gpurun_out/tune
mkdir -p $OUT
B="python bench.py --no-extras --cpu-sample 0 --steps 10 --warmup 2"
for v in prod r1shape onepass; do
  case $v in
    prod) E="";;
    r1shape) E="PCR_HIP_TUNE_SCATTER=3";;
    onepass) E="PCR_HIP_ONE_PASS=1";;
  esac
  env $E $B --workload C2 > $OUT/c2_$v.json 2> $OUT/c2_$v.err
  env $E $B --workload point_avg --grid 16384 --rows 2048 --points 125000000 --steps 4 > $OUT/c5s_$v.json 2> $OUT/c5s_$v.err
  env $E $B --workload gauss1 --steps 4 > $OUT/g1_$v.json 2> $OUT/g1_$v.err
done
PCR_HIP_TUNE_B=1 $B --workload C2 --steps 2 --warmup 1 > $OUT/c2_prof.json 2> $OUT/c2_prof.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/tune/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_step"].items() if not k.startswith("_")})
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h "cycles per block" $OUT/*.err | sort | uniq -c | sort -rn | head -3
