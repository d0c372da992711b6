# A/B of two builds of the library in alternating runs on one box:  bash tools/ab_lib.sh build_ab/libwesup_hip_old.so
OLD="$1"
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$1]', d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  run new || exit 1
  WESUP_HIP_LIB=$OLD run old || exit 1
done
