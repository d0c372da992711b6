# alternating bench.py runs of several flag sets on one box:  bash tools/ab_flags.sh 3 "--batch 1" "--batch 1 --no-step-plan" ...
N="$1"; shift
for i in $(seq $N); do
for f in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$f]', d['value'], 'img/s', d['ms_per_step'], d['ms_per_step_median'], d.get('step_plan'))" || exit 1
done; done
