"""The fused upsample + scatter-mean of the shallow layers (wesup_sp_pool_upsample_fwd, the segment form) alone on the GPU, per
layer shape of a training step, and the superpixel preprocessing.  (The tensors of a 4 x 480 x 480 step fit the 256 MB memory-side
cache: repeated launches over ONE tensor read from it -- the figure is an upper bound of what the step's launch sees.)  Microseconds per launch and GB/s on the layer's own bytes (the layer read once + labels + the pooled rows).

  python tools/pool_micro.py [--size 480] [--batch 4] [--grid 24] [--reps 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480)
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--grid', type=int, default=24)
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
d = torch.device('cuda:0')
B, S, g = args.batch, args.size, args.grid
import numpy as np
labs = np.stack([synth.voronoi_labels(3 + b, S, S, g) for b in range(B)])
masks = np.stack([synth.point_mask(3 + b, labs[b], 0.2, 2) for b in range(B)])
lab_d, mask_d = torch.from_numpy(labs).to(d), torch.from_numpy(masks).to(d)
Kmax = (g * g + 63) // 64 * 64


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


m = ops.sp_preprocess(lab_d, mask_d, Kmax)
us = timeit(lambda: ops.sp_preprocess(lab_d, mask_d, Kmax, into=m), args.reps)
print(f'# B={B} {S}x{S} g={g} Kmax={Kmax}')
print(f'sp_preprocess           {us:8.1f} us')
tot_old = 0.0
# the seven shallow layers of a 480 x 480 step with the side conv commuted behind the pooling: the conv outputs themselves
for name, div, C, n in [('conv1_x', 1, 64, 2), ('conv2_x', 2, 128, 2), ('conv3_x', 4, 256, 3)]:
    h = w = S // div
    if h * w <= 4096:
        continue
    s = torch.randn(B, h, w, C, device=d)
    out = torch.empty(B, Kmax, C, device=d)
    old = timeit(lambda: ops.sp_pool_upsample_fwd(s, m, out, 0), args.reps)
    by = 4.0 * B * (h * w * C + Kmax * C) + B * S * S
    tot_old += n * old
    print(f'{name} {h:4d}x{w:<4d} C={C:3d}: {old:7.1f} us {by / old * 1e-3:7.1f} GB/s  (x{n} per step)')
print(f'per step: {tot_old * 1e-3:.3f} ms')
