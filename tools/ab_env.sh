#!/bin/bash
# A/B of one environment knob on the SAME box, alternating runs: tools/ab_env.sh VAR "v1 v2 ..." REPEATS WORKLOAD [bench args]
# prints ms/step and the per-kernel table of each run (GPU boxes differ by a few per cent: never compare across calls)
var=$1; vals=$2; reps=$3; wl=$4; shift 4
for r in $(seq 1 $reps); do
  for v in $vals; do
    env $var=$v python3 bench.py --no-extras --cpu-sample 0 --workload $wl --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={a:b for a,b in d['kernels_ms_per_step'].items() if a!='_note'}
print('$var=$v', 'rep$r', d['ms_per_step'], k)"
  done
done
