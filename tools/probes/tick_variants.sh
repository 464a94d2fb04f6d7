# one line per variant: graph-replayed tick of 16 (us per tick) for CVGS_K1_RPW = default / 1 / 2 / 4 with the installed library
for RPW in default 1 2 4; do
  if [ $RPW = default ]; then unset CVGS_K1_RPW; else export CVGS_K1_RPW=$RPW; fi
  python tools/bench_tick.py 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rpw', os.environ.get('CVGS_K1_RPW','default'), 'tick16 graph us', d['graph']['us_per_tick'], 'eager', d['eager']['us_per_tick_wall'])"
done
unset CVGS_K1_RPW
