#!/bin/bash
# Builds the ablation variants of the headline kernel for tools/probes/tick_ablation.py (run in the container; build/ travels with the
# gpurun snapshot).  Each variant = k_k1_c3.hip (the translation unit that holds k1_resize_split<3, ..., SRC_U8, float>) compiled again
# with -DCVGS_K1_ABLATE=<bits> / -DCVGS_K1_STORE=<n> (csrc/k_k1_impl.hpp, csrc/k_taps.hpp), linked with the product's OTHER objects into
# build/ablate/libcvgs_<name>.so.  Nothing here goes into cvgpuspeedup_amd/lib/libcvgs_hip.so.
#   bash tools/probes/build_ablate.sh            # all variants
set -eu
cd "$(dirname "$0")/../.."
make -C cvgpuspeedup_amd/csrc -j8 >/dev/null
OBJ=build/csrc
OUT=build/ablate
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-command-line-argument -Wno-unused-variable -Wno-unused-but-set-variable -mllvm -amdgpu-kernarg-preload-count=14"
# name:ablate bits:store flavour
VARIANTS=${VARIANTS:-"full:0:0 ldst:2:0 ld:6:0 st:3:0 desc:16:0 zfast:8:0 zfast_ldst:10:0 plain:0:1 sc1:0:2 sys:0:3"}
# the product's other objects (the Makefile's SRCS; build/csrc may hold stale objects of removed sources)
SRCS=$(sed -n 's/^SRCS *= *//p' cvgpuspeedup_amd/csrc/Makefile)
others=""
for f in $SRCS; do [ "$f" = k_k1_c3.hip ] || others="$others $OBJ/$f.o"; done
build_one() {
  IFS=: read -r name bits st <<<"$1"
  /opt/rocm/bin/hipcc $FLAGS -DCVGS_K1_ABLATE=$bits -DCVGS_K1_STORE=$st -c cvgpuspeedup_amd/csrc/k_k1_c3.hip -o $OUT/k_k1_c3_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libcvgs_$name.so $OUT/k_k1_c3_$name.o $others -ldl
  rm -f $OUT/k_k1_c3_$name.o
  echo "built $OUT/libcvgs_$name.so"
}
export -f build_one
export FLAGS OUT others
printf '%s\n' $VARIANTS | xargs -P 4 -I{} bash -c 'build_one {}'
