#!/bin/bash
# Developer helper (measurement): build libeg_hip.so through an explicit device pipeline - device assembly, an edit of the listing
# (tools/strip_false_hazards.py), assembler, lld, offload bundle, host object with the bundle embedded - to measure what the wait states
# cost that the compiler puts after inline-asm statements (its hazard recogniser must assume an asm block may forward a partial register).
# usage: tools/build_stripped.sh NAME [strip|keep] [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; mode=${2:-strip}; shift; shift || true
LL=/opt/rocm/lib/llvm/bin
d=build_variants/$name; mkdir -p $d
for tu in eg_hip eg_gen; do
  (
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" --cuda-device-only -S -o $d/$tu.s elastic_elgamal_amd/csrc/$tu.hip 2>/dev/null
  if [ "$mode" = strip ]; then python3 tools/strip_false_hazards.py $d/$tu.s $d/$tu.edit.s; else cp $d/$tu.s $d/$tu.edit.s; fi
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $d/$tu.edit.s -o $d/$tu.dev.o
  $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $d/$tu.out $d/$tu.dev.o
  $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$d/$tu.out -output=$d/$tu.hipfb
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $d/$tu.hipfb \
      -c -o $d/$tu.o elastic_elgamal_amd/csrc/$tu.hip 2>/dev/null
  ) &
done
wait
hipcc --offload-arch=gfx950 -shared -o build_variants/libeg_$name.so $d/eg_hip.o $d/eg_gen.o
echo built build_variants/libeg_$name.so
