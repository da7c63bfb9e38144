#!/bin/bash
# A/B builds of the library with extra compiler flags:  bash tools/build_variant.sh <name> "<extra flags>"  ->  build/lib_<name>.so
# (the two translation units that do not include the kernels are compiled once into build/obj/)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC"
mkdir -p build/obj
for tu in bvh_builder_gpu det_splat; do
  if [ ! -f build/obj/$tu.o ] || [ clive2_amd/csrc/$tu.hip -nt build/obj/$tu.o ]; then
    hipcc $FLAGS -c clive2_amd/csrc/$tu.hip -o build/obj/$tu.o
  fi
done
hipcc $FLAGS $@ -c clive2_amd/csrc/renderer_api.hip -o build/obj/renderer_api_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/renderer_api_$NAME.o build/obj/bvh_builder_gpu.o build/obj/det_splat.o -o build/lib_$NAME.so
echo built build/lib_$NAME.so
