#!/bin/bash
# Builds library variants for tools/ab_libs.sh: tools/build_ab.sh TAG "-DFOO=1 -DBAR=2" [TAG2 "flags" ...]
# -> pointcloud-raster_amd/lib_ab/libpcr_hip_TAG.so (same sources, extra compiler flags; travels to the GPU box with the tree)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/pointcloud-raster_amd/csrc
mkdir -p $root/pointcloud-raster_amd/lib_ab
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  tmp=$(mktemp -d)
  for f in $src/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function \
      -I$root/include -I$src $flags -c $f -o $tmp/$(basename $f .hip).o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/pointcloud-raster_amd/lib_ab/libpcr_hip_$tag.so $tmp/*.o
  rm -rf $tmp
  echo built lib_ab/libpcr_hip_$tag.so "($flags)"
done
