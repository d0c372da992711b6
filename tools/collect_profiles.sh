#!/bin/bash
# Collects everything profiles/ holds for a round on the GPU box (one gpurun call):
#   bash tools/collect_profiles.sh r02
# 1. bench.py plain -> bench_line.json;  2. the same command under rocprofv3 --kernel-trace --stats -> kernel stats;
# 3. PMC counters over bench.py itself, one rocprofv3 run per counter group with no trace domain beside it
#    (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in passes of their own) -> traffic and MFMA-busy per kernel class;
# 4. the same two MFMA counters over tools/layer_pmc.py -> 13 layers x {fwd, dgrad, wgrad} table;
# 5. tools/layer_table.py (HIP-event times per layer and pass, direct kernels alone) and tools/wino_table.py (direct vs
#    Winograd-domain per layer);
# 6. tools/roofline_inputs.py turns 3+4 into profiles/rNN_roofline_inputs.json (read by bench.py) and rNN_layer_mfma.csv.
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# rocprofv3 starts the HIP runtime before python does: what bench.py sets at import time comes too late under the profiler
export GPU_MAX_HW_QUEUES=6
timeout -k 10 400 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
echo "stats done"
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
pmc() {   # name, counters, program args...
    local name=$1 ctrs=$2; shift 2
    timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -- python3 "$@" > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; return 1; }
    echo "pmc $name done"
}
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
pmc bench_fetch "FETCH_SIZE" $BENCH &&
pmc bench_write "WRITE_SIZE" $BENCH &&
pmc bench_mfma "SQ_VALU_MFMA_BUSY_CYCLES" $BENCH &&
pmc bench_busy "GRBM_GUI_ACTIVE" $BENCH &&
pmc layer_mfma "SQ_VALU_MFMA_BUSY_CYCLES" tools/layer_pmc.py $OUT/layer_manifest.json &&
pmc layer_busy "GRBM_GUI_ACTIVE" tools/layer_pmc.py $OUT/layer_manifest.json &&
pmc conv_lds "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" tools/conv_micro.py fwd 4 120 120 256 256 &&
pmc wgrad_lds "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" tools/conv_micro.py wgrad 4 120 120 256 256
timeout -k 10 200 python3 tools/layer_table.py 5 > $OUT/layer_table.txt 2>&1
timeout -k 10 200 python3 tools/wino_table.py 5 > $OUT/wino_table.txt 2>&1
python3 tools/pmc_summary.py $OUT/pmc_conv_lds $OUT/pmc_wgrad_lds > $OUT/pmc_lds_summary.csv
python3 tools/roofline_inputs.py $OUT $R > $OUT/roofline_inputs.log 2>&1 || { echo "roofline_inputs failed"; tail -5 $OUT/roofline_inputs.log; }
cp profiles/${R}_roofline_inputs.json profiles/${R}_layer_mfma.csv $OUT/ 2>/dev/null
# keep the merge-back small: drop the raw counter files
rm -rf $OUT/pmc_*/
ls -la $OUT
