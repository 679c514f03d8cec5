#!/bin/bash
# Runs ON THE GPU BOX: A/B of prebuilt library variants (pointcloud-raster_amd/lib_ab/libpcr_hip_<tag>.so), alternating runs in
# ONE call: tools/ab_libs.sh "tagA tagB ..." REPEATS WORKLOAD [bench args]
tags=$1; reps=$2; wl=$3; shift 3
for r in $(seq 1 $reps); do
  for t in $tags; do
    cp pointcloud-raster_amd/lib_ab/libpcr_hip_$t.so pointcloud-raster_amd/lib/libpcr_hip.so
    python3 bench.py --no-extras --cpu-sample 0 --workload $wl --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={a:b for a,b in d['kernels_ms_per_step'].items() if a!='_note'}
print('$t', 'rep$r', d['ms_per_step'], d['roofline']['avg_kernel_ms'], k)"
  done
done
