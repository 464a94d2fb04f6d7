cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/new.so
run() {
  for W in cfg3 nv12crops; do
    python tools/bench_more.py --iters 300 --only $W 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['config'][:60], j['us_per_launch'])"
  done
}
for rep in 1 2 3; do
  cp build/ab/libcvgs_hip_old.so cvgpuspeedup_amd/lib/libcvgs_hip.so; echo "OLD K4"; run
  cp /tmp/new.so cvgpuspeedup_amd/lib/libcvgs_hip.so; echo "NEW K4 (aspect-ratio window, default planes, RGB-order program)"; run
done
cp /tmp/new.so cvgpuspeedup_amd/lib/libcvgs_hip.so
