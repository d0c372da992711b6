# The second bench lines of a round (never the headline), one gpurun call:  bash tools/second_lines.sh r04
#   end-to-end (+ its rocprofv3 --stats), multi-scale rotations (12 shapes replayed / walked; 200 draws steady state / cold),
#   batch 1 (+ stats) and batch 2, walk vs replay at batch 4, wesup_slic alone.  Results under gpurun_out/lines/.
set -o pipefail
R=${1:-r04}
O=gpurun_out/lines; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=6
b() { local out=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $O/$out.json 2> $O/$out.err || { tail -3 $O/$out.err; exit 1; }; }
b ${R}_e2e_bench_line --end-to-end --steps 40 --warmup 10
b ${R}_multiscale_bench_line --multiscale 12 --no-kernel-timing --steps 120 --warmup 60
b ${R}_multiscale_walk_bench_line --multiscale 12 --no-kernel-timing --steps 120 --warmup 60 --no-step-plan
b ${R}_multiscale200_bench_line --multiscale 200 --no-kernel-timing --steps 200 --warmup 1000     # ~94 distinct shapes, steady state
b ${R}_multiscale200_cold_bench_line --multiscale 200 --no-kernel-timing --steps 200 --warmup 0   # every shape for the first time
b ${R}_b1_bench_line --batch 1 --steps 60 --warmup 10
b ${R}_b1_walk_bench_line --batch 1 --no-kernel-timing --steps 60 --warmup 10 --no-step-plan
b ${R}_b2_bench_line --batch 2 --steps 60 --warmup 10
b ${R}_b2_walk_bench_line --batch 2 --no-kernel-timing --steps 60 --warmup 10 --no-step-plan
b ${R}_b4_walk_bench_line --no-kernel-timing --steps 40 --no-step-plan
b ${R}_b4_replay_bench_line --no-kernel-timing --steps 40
for t in e2e b1; do
  extra="--batch 1"; [ $t = e2e ] && extra="--end-to-end"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$t -- python3 bench.py $extra --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 10 > $O/${t}_under_rocprof.json 2> $O/${t}_rocprof.err || { tail -3 $O/${t}_rocprof.err; exit 1; }
  find $O/st_$t -name "*kernel_stats.csv" -exec cp {} $O/${R}_${t}_kernel_stats.csv \;
  rm -rf $O/st_$t
done
timeout -k 10 120 python3 tools/slic_micro.py 2>&1 | grep -v amdgpu > $O/${R}_slic.txt
for f in $O/${R}_*bench_line.json; do python3 -c "import json; d=json.load(open('$f')); print('$(basename $f)', d['value'], d['ms_per_step'], d.get('step_plan'))"; done; cat $O/${R}_slic.txt
