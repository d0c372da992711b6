"""Summarise rocprofv3 --pmc runs: mean counter value per launch for every (kernel, counter).

  python tools/pmc_summary.py gpurun_out/pmc_* > profiles/rNN_pmc_summary.csv
Each directory is one rocprofv3 run (counters that cannot share a pass go to separate runs); the first launch of a
kernel in a run (warm-up) is dropped when there are more than two."""
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
