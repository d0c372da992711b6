"""Host time of the FIRST / SECOND / THIRD occurrence of N multi-scale shapes at batch 1 (the reference's operating point,
utils/data.py:98-101) without a profiler, under a garbage-collector setting:  python tools/cold_walk.py [N=40] [gc: on|off|freeze]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.data import LabelMaps
from wesup_amd.utils.metrics import accuracy, dice

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else 'on'
dev = torch.device('cuda:0')
t = initialize_trainer('wesup', device='cuda:0')
t.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
t.optimizer, _ = t.get_default_optimizer()
t.metric_funcs = [accuracy, dice]
t.model.train(); t.tracker.train()
t.kwargs['max_superpixels'] = None
pool = []
for i in range(N):
    f = 0.3 + 0.1 * i / N
    h, w = int(522 * f), int(775 * f)
    gi = max(2, int(round((h * w / 200.0) ** 0.5)))
    imgs, labs, pts, pix = synth.make_batch(i + 1, 1, h, w, gi)
    d = torch.from_numpy(labs).to(dev)
    pool.append((torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), LabelMaps(d, [int(d.max()) + 1])))
for _ in range(3):
    imgs, labs, pts, pix = synth.make_batch(999, 1, 150, 230, 13)
    d = torch.from_numpy(labs).to(dev)
    t.train_one_iteration('train', torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), LabelMaps(d, [int(d.max()) + 1]))
torch.cuda.synchronize()
if mode == 'off':
    gc.disable()
elif mode == 'freeze':
    gc.collect(); gc.freeze()
print(f'# gc {mode}; counts {gc.get_count()}, thresholds {gc.get_threshold()}, tracked objects {len(gc.get_objects())}')
for name in ('first', 'second', 'third (replay)', 'fourth (replay)'):
    torch.cuda.synchronize()
    g0 = [s['collections'] for s in gc.get_stats()]
    t0 = time.perf_counter()
    per = []
    for data in pool:
        a = time.perf_counter()
        t.train_one_iteration('train', *data)
        per.append(time.perf_counter() - a)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g1 = [s['collections'] for s in gc.get_stats()]
    per.sort()
    print(f'{name:16s}: {dt / N * 1e3:6.2f} ms per step (host only {host / N * 1e3:6.2f}; median {per[N // 2] * 1e3:5.2f}, max {per[-1] * 1e3:6.2f}); '
          f'collections gen0/1/2: {[b - a for a, b in zip(g0, g1)]}; {t.step_runner().stats}')
