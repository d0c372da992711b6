# MFMA || VALU co-execution probe (tools/probes/coexec.hip): timings, then the counters in a pass of their own.
#   gpurun -- 'bash tools/coexec.sh [--pmc]'      -> gpurun_out/coexec/{timings.txt,counters.csv}
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coexec; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probes/coexec.hip -o $O/coexec || exit 1
timeout -k 10 120 $O/coexec > $O/timings.txt 2>&1 || { echo probe failed; tail -5 $O/timings.txt; exit 1; }
cat $O/timings.txt
[ "$1" = "--pmc" ] || { rm -f $O/coexec; exit 0; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc -- $O/coexec > $O/pmc.log 2>&1 || { echo pmc failed; tail -5 $O/pmc.log; exit 1; }
python3 - $O/pmc > $O/counters.csv <<'PY'
import sys, os, glob, csv, collections
print('kernel,counter,mean_per_launch,launches')
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
        acc.setdefault(k, collections.OrderedDict()).setdefault(r['Dispatch_Id'], 0.0)
        acc[k][r['Dispatch_Id']] += float(r['Counter_Value'])
    for (kern, ctr), per in acc.items():
        vals = list(per.values())
        print(f'"{kern}",{ctr},{sum(vals) / len(vals):.1f},{len(vals)}')
PY
rm -rf $O/pmc $O/coexec
cat $O/counters.csv
