# alternating bench.py runs of an older checkout (git worktree add _old <commit>; make -C _old/wesup_amd/csrc) and the working tree on one box
for i in 1 2 3 4; do
  for d in _old .; do
    (cd $d && timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$d]', d['value'], 'img/s', d['ms_per_step'], d['ms_per_step_median'])") || exit 1
  done
done
