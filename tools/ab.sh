run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run
run --ddp-probe pg
run --force-ddp
done
