# alternating bench.py runs of several settings on one box; a setting is "[VAR=value ...] [bench flags]":
#   bash tools/ab_flags.sh 3 "--batch 1" "--batch 1 --no-step-plan" "GPU_MAX_HW_QUEUES=4 --batch 1"
N="$1"; shift
for i in $(seq $N); do
for f in "$@"; do
  envs=(); flags=()
  for w in $f; do if [[ "$w" == [A-Z_]*=* && ${#flags[@]} -eq 0 ]]; then envs+=("$w"); else flags+=("$w"); fi; done
  timeout -k 10 300 env "${envs[@]}" python bench.py --no-cpu-baseline --no-kernel-timing "${flags[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$f]', d['value'], 'img/s', d['ms_per_step'], d['ms_per_step_median'], d.get('step_plan'))" || exit 1
done; done
