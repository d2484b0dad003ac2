#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: build_variants/librt_<name>.so, the HIP library with extra definitions (A/B runs: RT_AMD_LIB, tools/ab_scene.sh)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p build_variants
C=ray-tracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wno-unused-result \
  "-DRT_TABLES_PATH=\"$PWD/ray-tracer_amd/data/tables.bin\"" "$@" -o build_variants/librt_$name.so $C/rt_kernels.hip $C/rt_flatten.cpp $C/rt_tables.cpp $C/rt_host_util.cpp
