"""Mean FETCH_SIZE (KiB past L2) per gemm_tn_kernel launch class from a rocprofv3 --pmc FETCH_SIZE pass:
  python tools/tn_fetch.py gpurun_out/pmc_dir"""
import collections, csv, glob, os, re, sys
acc = collections.OrderedDict()
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE':
            continue
        k = (int(r['Dispatch_Id']), r['Kernel_Name'])
        acc[k] = acc.get(k, 0.0) + float(r['Counter_Value'])
tot, n = collections.Counter(), collections.Counter()
for (_, kern), v in acc.items():
    m = re.search(r'gemm_tn_kernel<([^>]*)>', kern)
    if m:
        tot[m.group(1)] += v
        n[m.group(1)] += 1
for k in tot:
    print(f'gemm_tn_kernel<{k}>: {n[k]} launches, FETCH {tot[k] / n[k] / 1024:.1f} MiB per launch')
