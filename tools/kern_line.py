"""Prints the per-class kernel table of a bench line (stdin): in-step and alone ms per step, rate."""
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'roofline', d['roofline']['frac'], d['roofline']['achieved'])
tot_alone = 0.0
for k, v in d['kernels'].items():
    if k.startswith('_'):
        if k != '_note': print(k, v)
        continue
    tot_alone += v.get('alone_ms_per_step', 0.0)
    print(f"{k:20s} in-step {v['ms_per_step']:7.3f} ms  alone {v.get('alone_ms_per_step', float('nan')):7.3f} ms  launches {v['launches_per_step']:5.1f}  "
          + (f"{v['tflops']:6.1f} TF" if 'tflops' in v else f"{v.get('alone_gbs', v.get('gbs', 0)):7.1f} GB/s alone"))
print('sum of alone times', round(tot_alone, 3))
