#!/bin/bash
# tools/wino_variant_build.sh <name> -DFLAG ...: dl4ds_amd/libdl4ds_hip_<name>.so with the Winograd translation units rebuilt under extra
# flags (load it with DL4DS_HIP_LIB; *.so is git-ignored and travels to the GPU box, gpurun_variants/ does not)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; shift
mkdir -p $R/gpurun_variants/obj_$NAME
OBJS=""
for f in $R/dl4ds_amd/csrc/*.hip $R/dl4ds_amd/csrc/*.cpp; do b=$(basename $f); case $b in conv_wino*.hip) continue;; esac; OBJS="$OBJS $R/dl4ds_amd/csrc/_build/$b.o"; done
for k in _22 _23 _32 _33 _wgrad "" 4_22 4_32; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -x hip "$@" -c $R/dl4ds_amd/csrc/conv_wino$k.hip -o $R/gpurun_variants/obj_$NAME/conv_wino$k.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dl4ds_amd/libdl4ds_hip_$NAME.so $OBJS $R/gpurun_variants/obj_$NAME/conv_wino*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built dl4ds_amd/libdl4ds_hip_$NAME.so
