"""Host-side profile of the Python walk of a training iteration at the reference's operating point (batch 1, a new
0.3 - 0.4 x (775 x 522) shape per step, models/wesup.py:178, utils/data.py:98-101): cProfile over the FIRST occurrence of N
shapes (a recording walk each) and over their SECOND (recording + diff), top functions by own time.

  python tools/walk_profile.py [N=40]
"""
import cProfile
import os
import pstats
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
dev = torch.device('cuda:0')
t = initialize_trainer('wesup', device='cuda:0')
t.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
t.optimizer, _ = t.get_default_optimizer()
t.metric_funcs = [accuracy, dice]
t.model.train(); t.tracker.train()
t.kwargs['max_superpixels'] = None
rs = np.random.RandomState(7)
pool = []
for i in range(N):
    f = 0.3 + 0.1 * i / N
    h, w = int(522 * f), int(775 * f)
    gi = max(2, int(round((h * w / 200.0) ** 0.5)))
    imgs, labs, pts, pix = synth.make_batch(i + 1, 1, h, w, gi)
    d = torch.from_numpy(labs).to(dev)
    pool.append((torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), LabelMaps(d, [int(d.max()) + 1])))
for _ in range(3):                       # the run's first iterations (optimiser's first step, library warm-up) on a shape of their own
    imgs, labs, pts, pix = synth.make_batch(999, 1, 150, 230, 13)
    d = torch.from_numpy(labs).to(dev)
    t.train_one_iteration('train', torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), LabelMaps(d, [int(d.max()) + 1]))
torch.cuda.synchronize()
for name in ('first occurrence', 'second occurrence', 'third occurrence (replay)'):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    for data in pool:
        t.train_one_iteration('train', *data)
    pr.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'==== {name}: {dt / N * 1e3:.2f} ms per step (with the profiler on), {t.step_runner().stats}')
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    st.print_stats(22)
