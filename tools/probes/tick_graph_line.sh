python tools/bench_tick.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tick16 graph us', d['graph']['us_per_tick'], 'p10/p90', d['graph']['p10_us'], d['graph']['p90_us'], 'eager', d['eager']['us_per_tick_wall'], 'ok', d['bit_identical_to_cvgs_execute'])"
