#!/bin/bash
# Runs ON THE GPU BOX: the Gaussian sigma = 1 leg on the window shapes of a C5 shard (N = 8 / 4 / 2), default tile plan against
# PCR_HIP_CELL_TILE_H=20 (20-row cell tiles, as many row bands as that takes).
mkdir -p gpurun_out/shardg
for shape in "2048 125000000" "4096 250000000" "8192 500000000"; do
  set -- $shape
  for th in default 20; do
    if [ $th = default ]; then unset PCR_HIP_CELL_TILE_H; else export PCR_HIP_CELL_TILE_H=$th; fi
    python3 bench.py --grid 16384 --rows $1 --points $2 --workload gauss1 --no-extras --cpu-sample 0 --steps 3 --warmup 2 2>gpurun_out/shardg/err_$1_$th.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={a:b for a,b in d['kernels_ms_per_step'].items() if a!='_note'}
print('rows $1 tile_h $th:', d['ms_per_step'], 'ms', d['config']['num_bins'], 'bins', d['config']['lds_tile'], k)"
  done
done
