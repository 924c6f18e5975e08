#!/bin/bash
# Build a variant of libdl4ds_hip.so with one source recompiled under extra -D flags:
#   tools/variant_build.sh <name> <source.hip> -DFOO=1 ...   -> dl4ds_amd/libdl4ds_hip_<name>.so
# Run with DL4DS_HIP_LIB=.../dl4ds_amd/libdl4ds_hip_<name>.so (*.so is git-ignored and travels to the GPU box; gpurun_variants/ holds only the objects).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p $R/gpurun_variants/obj_$NAME
python $R/dl4ds_amd/csrc/build.py > /dev/null
OBJS=""
for f in $R/dl4ds_amd/csrc/*.hip $R/dl4ds_amd/csrc/*.cpp; do
  b=$(basename $f)
  if [ "$b" == "$SRC" ]; then continue; fi
  OBJS="$OBJS $R/dl4ds_amd/csrc/_build/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -x hip "$@" -c $R/dl4ds_amd/csrc/$SRC -o $R/gpurun_variants/obj_$NAME/$SRC.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dl4ds_amd/libdl4ds_hip_$NAME.so $OBJS $R/gpurun_variants/obj_$NAME/$SRC.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built dl4ds_amd/libdl4ds_hip_$NAME.so
