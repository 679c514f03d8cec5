for t in base big; do
  cp pointcloud-raster_amd/lib_ab/libpcr_hip_$t.so pointcloud-raster_amd/lib/libpcr_hip.so
  for tl in 1 0; do
    PCR_HIP_DEBUG_TWO_LEVEL=$tl python3 bench.py --grid 16384 --rows 8192 --points 500000000 --workload point_avg --no-extras --cpu-sample 0 --steps 5 --warmup 2 2>gpurun_out/r05w/err_${t}_$tl.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={a:b for a,b in d['kernels_ms_per_step'].items() if a!='_note'}
print('$t two_level=$tl', d['ms_per_step'], d['config']['num_bins'], k)"
  done
done
