# kernel-argument preload of K1's leading scalars: the launch-bound regimes (run on the GPU box; compare trees / PRELOAD= builds)
python bench.py --no-cpu --no-extra --no-regimes --no-sweep --no-queue-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ticks16 graph us/frame', d['ms_per_step']*1e3, 'one_launch graph', d.get('one_launch_per_step'), 'lat', d.get('latency_us'), 'tick64', d.get('tick64_us_per_step'))"
python bench.py --submission graph --eager --steps 256 --warmup 64 --no-cpu --no-extra --no-regimes --no-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('one launch per frame EAGER us/frame', d['ms_per_step']*1e3)"
python tools/bench_tick.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tick16 graph', d['graph'], 'eager', d['eager']['us_per_tick_wall'], 'eager_producer', d['eager_producer']['us_per_tick_wall'])"
