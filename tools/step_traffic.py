"""Traffic past L2 of one whole training step, per kernel, from rocprofv3 --pmc passes over bench.py:
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
  rocprofv3 --pmc WRITE_SIZE ... -d OUT/w ...
  python tools/step_traffic.py OUT/f OUT/w STEPS|auto
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE under-reports wide streaming reads 2x)."""
import collections, csv, glob, os, re, sys


def read(d, ctr):
    acc = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr:
                k = (int(r['Dispatch_Id']), r['Kernel_Name'])
                acc[k] = acc.get(k, 0.0) + float(r['Counter_Value'])
    out = collections.Counter()
    n = collections.Counter()
    for (_, kern), v in acc.items():
        name = re.sub(r'^void ', '', kern).split('(')[0]
        out[name] += v
        n[name] += 1
    return out, n


f, nf = read(sys.argv[1], 'FETCH_SIZE')
w, _ = read(sys.argv[2], 'WRITE_SIZE')
# steps: given, or 'auto' = the number of prop_kernel launches in the run (one per training step)
# (round 5: the optimiser step is two launches per step except the run's first, which is not split -- prop_kernel runs once per step)
steps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != 'auto' else (int(nf.get('prop_kernel', 0)) or int(nf.get('sgd_kernel', 0)) or 3)
rows = sorted(((2 * f[k] + w[k]) * 1024 / steps, k) for k in set(f) | set(w))
# kernels that are not part of the step (bench.py's roofline leg materialises the feature map once and pools it)
extra = [r for r in rows if r[1].startswith(('sp_pool_fwd_kernel', 'upsample_fwd_kernel'))]
rows = [r for r in rows if r not in extra]
tot = sum(r[0] for r in rows)
keep = {k for _, k in rows}
print(f'total {(tot) / 1e9:.2f} GB per step over {steps} steps (fetch x2 {sum(f[k] for k in keep) * 2048 / steps / 1e9:.2f}, '
      f'write {sum(w[k] for k in keep) * 1024 / steps / 1e9:.2f}); not counted: {", ".join(sorted(k for _, k in extra)) or "-"}')
for b, k in reversed(rows[-28:]):
    print(f'{b / 1e6:10.1f} MB/step  {nf[k] / steps:6.1f} launches  {k[:110]}')
