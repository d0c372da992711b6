# A/B of an environment setting in alternating bench.py runs on one box:  bash tools/ab_env.sh WESUP_WINO_FUSED_SHAPE=auto [bench flags]
SET="$1"; shift
run() { timeout -k 10 200 env $2 python bench.py --no-cpu-baseline --no-kernel-timing "${@:3}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$1]', d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  run base "X_=1" "$@" || exit 1
  run "$SET" "$SET" "$@" || exit 1
done
