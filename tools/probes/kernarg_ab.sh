mkdir -p gpurun_out/r05_x
for V in unset 0 1; do
  if [ $V = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$V; fi
  echo "== HIP_FORCE_DEV_KERNARG=$V"
  python bench.py --no-cpu --no-extra --no-regimes --no-sweep --no-queue-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ticks16 graph us/frame', d['ms_per_step']*1e3, 'one_launch graph', d.get('one_launch_per_step'), 'lat', d.get('latency_us'))"
  python bench.py --submission graph --eager --steps 256 --warmup 64 --no-cpu --no-extra --no-regimes --no-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('one launch per frame EAGER us/frame', d['ms_per_step']*1e3)"
  python tools/bench_tick.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tick16 eager', d['eager'], 'eager_producer', d['eager_producer'], 'host 1-crop', d.get('host_us_per_call_1_crop_chains'))"
done
