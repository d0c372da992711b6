# A/B of two bench.py flag sets in alternating runs on one box:  bash tools/ab.sh "" "--no-dual-transform"
A="$1"; Bf="$2"
for i in 1 2 3; do
for f in "$A" "$Bf"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$f]', d['ms_per_step'], d['ms_per_step_median'])" || exit 1
done; done
