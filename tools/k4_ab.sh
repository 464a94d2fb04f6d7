#!/bin/bash
# A/B of K4 (NV12 resize) build variants under build/ab/ on the GPU box, with the surfaces rotated so that the Infinity Cache
# cannot hold them (tools/bench_more.py): swaps the product library in place, runs cfg #3 and the 50-crop NV12 batch, restores
# the library.  Variants (all built with -DCVGS_K4_AB_RPW, which makes rows-per-wave selectable through CVGS_K4_RPW):
#   2: production settings (4 waves per workgroup)   3: 2 waves per workgroup   4: 8 waves per workgroup   5: chroma-row skip
# Build the variants first (here, before gpurun: build/ travels with the snapshot):
#   cd cvgpuspeedup_amd/csrc; mkdir -p ../../build/ab; OBJS=$(ls ../../build/csrc/*.o | grep -v "k_nv12.hip.o\|exp")
#   for V in "2:" "3:-DCVGS_K4_WPB=2" "4:-DCVGS_K4_WPB=8" "5:-DCVGS_K4_UVSKIP"; do N=${V%%:*}; F=${V#*:}
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCVGS_K4_AB_RPW $F -c k_nv12.hip -o ../../build/ab/k_nv12_$N.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libcvgs_hip_$N.so $OBJS ../../build/ab/k_nv12_$N.o -ldl; done
# Results: profiles/r02_i_k4_ab_hbm.txt.
cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/orig.so
run() {
  for W in cfg3 nv12crops; do
    python tools/bench_more.py --iters 300 --only $W 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['config'][:44], j['us_per_launch'])"
  done
}
for rep in 1 2; do
  cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so; echo "product library"; run
  for V in 2 3 4 5; do
    cp build/ab/libcvgs_hip_$V.so cvgpuspeedup_amd/lib/libcvgs_hip.so
    for R in 1 2 4; do
      [ $V != 2 ] && [ $R = 4 ] && continue
      echo "variant $V rows_per_wave=$R"
      CVGS_K4_RPW=$R run
    done
  done
done
cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so
