#!/bin/bash
# Kernel experiments: build libr2s_hip variants with extra -D flags into scratch/variants/ (git-ignored, travels with gpurun).
#   tools/profiling/build_variants.sh name "flags" [name "flags" ...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/real2sim-eval_amd/csrc
mkdir -p $R/scratch/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=$R/scratch/variants/obj_$name; mkdir -p $d
  for f in common raster physics skinning metrics robot_gs obs camera; do
    # only physics.hip / raster.hip see the experiment flags; the rest is linked from the product build
    if [ $f = raster ] || { [ $f = physics ] && [ -z "$RASTER_ONLY" ]; }; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $flags -c $C/$f.hip -o $d/$f.o &
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/common.o $d/raster.o $( [ -z "$RASTER_ONLY" ] && echo $d/physics.o || echo $C/physics.o ) $C/skinning.o $C/metrics.o $C/robot_gs.o $C/obs.o $C/camera.o \
      -o $R/scratch/variants/libr2s_$name.so -Wl,-rpath,/opt/rocm/lib
  echo built $name: $flags
done
