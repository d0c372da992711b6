#!/bin/bash
# Collects everything profiles/ holds for a round on the GPU box (one gpurun call):
#   bash tools/collect_profiles.sh r01
# 1. bench.py plain -> bench_line.json;  2. the same command under rocprofv3 --kernel-trace --stats -> kernel stats;
# 3. PMC counters of the dominant kernels (conv forward, conv wgrad, scatter-mean), one rocprofv3 run per counter
#    group with no trace domain beside it (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in passes of their own).
set -o pipefail
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
echo "stats done"
pmc() {   # name, counters, program args...
    local name=$1 ctrs=$2; shift 2
    timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -- python3 "$@" > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; return 1; }
    echo "pmc $name done"
}
CONV="tools/conv_micro.py fwd 4 120 120 256 256"
WGR="tools/conv_micro.py wgrad 4 120 120 256 256"
pmc conv_mfma "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" $CONV &&
pmc conv_wait "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" $CONV &&
pmc conv_lds "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" $CONV &&
pmc conv_fetch "FETCH_SIZE" $CONV &&
pmc conv_write "WRITE_SIZE" $CONV &&
pmc wgrad_mfma "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" $WGR &&
pmc wgrad_lds "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" $WGR &&
pmc pool_fetch "FETCH_SIZE" tools/pool_micro.py &&
pmc pool_write "WRITE_SIZE" tools/pool_micro.py
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary.csv
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge-back small: drop the raw traces
rm -rf $OUT/stats $OUT/pmc_*/
ls -la $OUT
