# quick second lines on one box, alternating settings (the cold multi-scale line is host-bound: it moves with the box's host load)
O=gpurun_out/lines2; mkdir -p $O
b() { local out=$1; shift; timeout -k 10 300 env $ENVV python3 bench.py --no-cpu-baseline "$@" > $O/$out.json 2> $O/$out.err || { tail -3 $O/$out.err; exit 1; }; python3 -c "import json; d=json.load(open('$O/$out.json')); print('$out', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d.get('step_plan'))"; }
for i in 1 2 3; do
ENVV="WESUP_PLAN_TRUST=1" b ms200_cold --multiscale 200 --no-kernel-timing --steps 200 --warmup 0
ENVV="WESUP_PLAN_TRUST=0" b ms200_cold_twin --multiscale 200 --no-kernel-timing --steps 200 --warmup 0
done
