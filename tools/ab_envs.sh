# alternating bench.py runs under several environment settings on one box:  bash tools/ab_envs.sh 2 "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=6" -- [bench flags]
N="$1"; shift
SETS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do SETS+=("$1"); shift; done
shift
for i in $(seq $N); do
for e in "${SETS[@]}"; do
  timeout -k 10 300 env $e python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], 'img/s', d['ms_per_step'], d['ms_per_step_median'])" || exit 1
done; done
