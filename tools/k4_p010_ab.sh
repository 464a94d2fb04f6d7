# A/B of the K4 kernel before / after the P010 (16-bit tap) template parameter: build/ab/libcvgs_hip_1.so holds the previous
# k_nv12.hip (P010 then falls to the interpreted kernel); same box, three alternations.
cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/orig.so
for rep in 1 2 3; do
for V in 0 1; do
  if [ $V = 0 ]; then cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so; else cp build/ab/libcvgs_hip_$V.so cvgpuspeedup_amd/lib/libcvgs_hip.so; fi
  echo "variant $V (0 = new K4 with the S16 template, 1 = previous source)"
  for W in cfg3 nv12crops; do
    python tools/bench_more.py --iters 400 --only $W 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['config'][:44], j['us_per_launch'])"
  done
done
done
cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so
