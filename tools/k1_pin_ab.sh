#!/bin/bash
# A/B of K1 with and without the SGPR-pinned row pointers (csrc/k_taps.hpp pin_uniform).  Build the variant first:
#   cd cvgpuspeedup_amd/csrc && for f in k_k1 k_k1_c3 k_k1_c4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCVGS_NO_PIN -c $f.hip -o ../../build/ab/${f}_nopin.o; done
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libcvgs_hip_1.so $(ls ../../build/csrc/*.o | grep -v "k_k1\|exp") ../../build/ab/k_k1*_nopin.o -ldl
# Round 2 result: no measurable difference on K1 (headline 4.437 vs 4.434 us, 16 x 50 crops 38.30 vs 38.42 us, whole-frame
# resizes within noise); the pin is kept because K4 gains from it (cfg #3 8.02 vs 8.45 us, tools/k4_ab.sh).
cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/o.so
for V in 0 1; do
  [ $V = 0 ] || cp build/ab/libcvgs_hip_$V.so cvgpuspeedup_amd/lib/libcvgs_hip.so
  echo "variant $V ($([ $V = 0 ] && echo pinned || echo no-pin))"
  python tools/bench_more.py --iters 200 --only cfg4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['config'][40:110], j['us_per_update'])"
  python bench.py --steps 256 --no-cpu --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('    headline', j['ms_per_step']*1000)"
  python bench.py --frames-per-launch 16 --steps 64 --no-cpu --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('    m16', j['ms_per_step']*1000)"
  python tools/bench_resize.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['case'], j['us'])" | head -9
done
cp /tmp/o.so cvgpuspeedup_amd/lib/libcvgs_hip.so
