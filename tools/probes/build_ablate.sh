#!/bin/bash
# Builds the ablation variants of the headline kernels for tools/probes/tick_ablation.py (run in the container; build/ travels with the
# gpurun snapshot).  Each variant = ONE translation unit compiled again with ablation macros (csrc/k_k1_impl.hpp, csrc/k_taps.hpp:
# CVGS_K1_ABLATE / CVGS_K1_STORE; csrc/k_nv12_x2.hip: CVGS_K4_ABLATE; csrc/k_k1_x4.hip: CVGS_X4_ABLATE), linked with the product's OTHER objects into
# build/ablate/libcvgs_<name>.so.  Nothing here goes into cvgpuspeedup_amd/lib/libcvgs_hip.so.
#   bash tools/probes/build_ablate.sh            # all variants
#   VARIANTS="k4_full:k_nv12_x2.hip:-DCVGS_K4_ABLATE=0" bash tools/probes/build_ablate.sh
set -eu
cd "$(dirname "$0")/../.."
make -C cvgpuspeedup_amd/csrc -j8 >/dev/null
OBJ=build/csrc
OUT=build/ablate
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-command-line-argument -Wno-unused-variable -Wno-unused-but-set-variable -mllvm -amdgpu-kernarg-preload-count=14"
# name:translation unit:macros (comma separated)
K1=k_k1_c3.hip
K4=k_nv12_x2.hip
X4=k_k1_x4.hip
# the default set is what tools/profile_r06_final.sh measures; other shapes on request, e.g.
#   VARIANTS="zfast:$K1:-DCVGS_K1_ABLATE=8 sc1:$K1:-DCVGS_K1_STORE=2 wpb4:$K1:-DCVGS_K1_WPB=4 xcdwl:$K1:-DCVGS_K1_ABLATE=32 k4_rows4w4:$K4:-DCVGS_K4_ROWS=4,-DCVGS_K4_WAVES=4 x4_now16:$X4:-DCVGS_X4_NO_W16=1 ldsys:$K1:-DCVGS_K1_LDSCOPE=3"
VARIANTS=${VARIANTS:-"full:$K1:-DCVGS_K1_ABLATE=0 ldst:$K1:-DCVGS_K1_ABLATE=2 ld:$K1:-DCVGS_K1_ABLATE=6 st:$K1:-DCVGS_K1_ABLATE=3 desc:$K1:-DCVGS_K1_ABLATE=16 \
k4_full:$K4:-DCVGS_K4_ABLATE=0 k4_ldst:$K4:-DCVGS_K4_ABLATE=2 k4_ld:$K4:-DCVGS_K4_ABLATE=6 k4_st:$K4:-DCVGS_K4_ABLATE=3 k4_math:$K4:-DCVGS_K4_ABLATE=5 k4_empty:$K4:-DCVGS_K4_ABLATE=16 \
x4_full:$X4:-DCVGS_X4_ABLATE=0 x4_ldst:$X4:-DCVGS_X4_ABLATE=2 x4_ld:$X4:-DCVGS_X4_ABLATE=6 x4_st:$X4:-DCVGS_X4_ABLATE=3 x4_math:$X4:-DCVGS_X4_ABLATE=5"}
SRCS=$(sed -n 's/^SRCS *= *//p' cvgpuspeedup_amd/csrc/Makefile)
build_one() {
  IFS=: read -r name tu defs <<<"$1"
  # the product's other objects (the Makefile's SRCS; build/csrc may hold stale objects of removed sources)
  others=""
  for f in $SRCS; do [ "$f" = "$tu" ] || others="$others $OBJ/$f.o"; done
  /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -c cvgpuspeedup_amd/csrc/$tu -o $OUT/tu_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libcvgs_$name.so $OUT/tu_$name.o $others -ldl
  rm -f $OUT/tu_$name.o
  echo "built $OUT/libcvgs_$name.so"
}
export -f build_one
export FLAGS OUT OBJ SRCS
printf '%s\n' $VARIANTS | xargs -P 4 -I{} bash -c 'build_one {}'
