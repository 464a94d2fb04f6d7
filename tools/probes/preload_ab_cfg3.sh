# cfg #3 (one NV12 6K surface per launch) with and without kernel-argument preload of the surface's descriptor (same box)
python tools/bench_more.py --only cfg3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print({k:d[k] for k in d if k in ('config','us_per_launch','us_per_frame','frac_of_8TBs','kernel','per_launch','queue','submission')})"
