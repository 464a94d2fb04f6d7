for R in 32 64 128; do for G in 511 767; do
echo "rows $R G $G"; CVGS_QUEUE_DEEP_ROWS=$R CVGS_QUEUE_G=$G ./tools/probes/bin_shc | grep -E "batch  (50|70)"; CVGS_QUEUE_DEEP_ROWS=$R CVGS_QUEUE_G=$G python bench.py --no-cpu --no-extra | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench us/step', j['ms_per_step']*1e3)"
done; done
