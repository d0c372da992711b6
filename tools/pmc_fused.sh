# Counters of the one-kernel Winograd product route on conv1_2 (profiles/rNN_pmc_fused_k64.csv), one rocprofv3 run per counter group:
#   gpurun -- 'bash tools/pmc_fused.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_fused; mkdir -p $O
run() { n=$1; c=$2; timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/$n -- python3 tools/fused_micro.py --only conv1_2 --reps 3 > $O/$n.log 2>&1 || { echo fail $n; tail -3 $O/$n.log; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
run b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
run c "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC"
run d "GRBM_GUI_ACTIVE"
# mean counter value per launch for every (kernel, counter) of the four runs; a kernel's first launch in a run (warm-up) is dropped
python3 - $O/a $O/b $O/c $O/d > $O/summary.csv <<'PY'
import sys, os, glob, csv, collections

print('run,kernel,counter,mean_per_launch,launches')
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
            acc.setdefault(k, collections.OrderedDict()).setdefault(r['Dispatch_Id'], 0.0)
            acc[k][r['Dispatch_Id']] += float(r['Counter_Value'])
        for (kern, ctr), per in acc.items():
            vals = list(per.values())
            if len(vals) > 2:
                vals = vals[1:]
            print(f'{os.path.basename(d.rstrip("/"))},"{kern}",{ctr},{sum(vals) / len(vals):.1f},{len(vals)}')
PY
rm -rf $O/a $O/b $O/c $O/d
cat $O/summary.csv
