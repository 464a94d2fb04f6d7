python tools/bench_more.py --only nv12crops 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print(d['config'][:60], d.get('us_per_launch'))"
python tools/bench_more.py --only nv12many 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print(d['config'][:70], {k:v for k,v in d.items() if k.startswith('us_')})"
