#!/bin/bash
# Collects everything profiles/ holds for one benchmark shape on the GPU box (one gpurun call):
#   bash tools/collect_profiles.sh r03                       # BASELINE configs[1]: batch 4, 480x480, 576 superpixels
#   bash tools/collect_profiles.sh r03_c4 4 800 39           # per-GPU shard of configs[3]
#   bash tools/collect_profiles.sh r03_c5 8 1024 55          # per-GPU shard of configs[4]
# 1. bench.py plain -> bench_line.json (run LAST, so that it reads this round's counter inputs);  2. the same command under
#    rocprofv3 --kernel-trace --stats -> kernel stats;
# 3. PMC counters over bench.py itself, one rocprofv3 run per counter group with no trace domain beside it
#    (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in passes of their own) -> traffic and MFMA-busy per kernel class;
# 4. (480 shape only) the same two MFMA counters over tools/layer_pmc.py -> 13 layers x {fwd, dgrad, wgrad} table, and
#    tools/layer_table.py (HIP-event times per layer and pass, direct kernels alone);
# 5. tools/wino_table.py at the shape (direct vs Winograd F(2x2) vs F(4x4) per layer and pass);
# 6. tools/roofline_inputs.py turns 3+4 into profiles/<tag>_roofline_inputs.json (read by bench.py) and <tag>_layer_mfma.csv;
# 7. tools/step_traffic.py: bytes past L2 per kernel and step from the two traffic passes of 3.
set -o pipefail
R=${1:-r03}
BATCH=${2:-4}; SIZE=${3:-480}; GRID=${4:-24}
SHAPE="--batch $BATCH --size $SIZE --grid $GRID"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# rocprofv3 starts the HIP runtime before python does: what bench.py sets at import time comes too late under the profiler
export GPU_MAX_HW_QUEUES=6
CPUB=""; [ "$SIZE" != "480" ] && CPUB="--no-cpu-baseline"      # the CPU baseline is config c1's: timed once, on the 480 line
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $SHAPE --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
echo "stats done"
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
pmc() {   # name, counters, program args...
    local name=$1 ctrs=$2; shift 2
    timeout -k 10 400 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -- python3 "$@" > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; return 1; }
    echo "pmc $name done"
}
BENCH="bench.py $SHAPE --steps 2 --warmup 1 --no-cpu-baseline"
pmc bench_fetch "FETCH_SIZE" $BENCH &&
pmc bench_write "WRITE_SIZE" $BENCH &&
pmc bench_mfma "SQ_VALU_MFMA_BUSY_CYCLES" $BENCH &&
pmc bench_busy "GRBM_GUI_ACTIVE" $BENCH
if [ "$SIZE" = "480" ]; then
    pmc layer_mfma "SQ_VALU_MFMA_BUSY_CYCLES" tools/layer_pmc.py $OUT/layer_manifest.json &&
    pmc layer_busy "GRBM_GUI_ACTIVE" tools/layer_pmc.py $OUT/layer_manifest.json
    timeout -k 10 200 python3 tools/layer_table.py 5 > $OUT/layer_table.txt 2>&1
fi
timeout -k 10 400 python3 tools/wino_table.py --size $SIZE --batch $BATCH --reps 3 > $OUT/wino_table.txt 2>&1
python3 tools/roofline_inputs.py $OUT $R $BATCH $SIZE $GRID > $OUT/roofline_inputs.log 2>&1 || { echo "roofline_inputs failed"; tail -5 $OUT/roofline_inputs.log; }
cp profiles/${R}_roofline_inputs.json profiles/${R}_layer_mfma.csv $OUT/ 2>/dev/null
# the bench line itself, last: the counter inputs of this shape and round exist now (its roofline.traffic reads them)
timeout -k 10 500 python3 bench.py $SHAPE $CPUB > $OUT/bench_line.json 2> $OUT/bench_line.err || { tail -5 $OUT/bench_line.err; exit 1; }
echo "bench done"
# traffic past L2 per kernel over a whole step (the same two counter passes, all kernels): profiles/<tag>_step_traffic.txt
python3 tools/step_traffic.py $OUT/pmc_bench_fetch $OUT/pmc_bench_write auto > $OUT/step_traffic.txt 2>/dev/null
# keep the merge-back small: drop the raw counter files
rm -rf $OUT/pmc_*/
ls -la $OUT
