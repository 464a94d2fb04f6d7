# eager ticks of 16 x 50 host-described crops: descriptors in the kernel arguments (default) against the stream's pinned table ring
# (CVGS_MANY_INLINE=0), alternating on one box
for R in 1 2 3; do
  for V in 1 0; do
    CVGS_MANY_INLINE=$V python tools/bench_tick.py 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('inline=$V', 'graph', d['graph']['us_per_tick'], 'eager dev tables', d['eager_device_tables']['us_per_tick_wall'], 'eager host desc', d['eager']['us_per_tick_wall'], 'with producer', d['eager_producer']['us_per_tick_wall'], 'host per call (8x8 target)', d.get('host_us_per_call_50_crop_chains_8x8_target'), d['bit_identical_to_cvgs_execute'])"
  done
done
