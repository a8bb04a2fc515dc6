#!/bin/bash
# Developer helper: build an alternate libeg_hip.so with extra -D flags for A/B runs (EG_LIB=build_variants/libeg_NAME.so).
# usage: tools/build_variant.sh NAME [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants/$name
for tu in eg_hip eg_gen; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c -o build_variants/$name/$tu.o elastic_elgamal_amd/csrc/$tu.hip &
done
wait
hipcc --offload-arch=gfx950 -shared -o build_variants/libeg_$name.so build_variants/$name/eg_hip.o build_variants/$name/eg_gen.o
echo built build_variants/libeg_$name.so
