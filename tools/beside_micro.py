"""How much of a memory-bound side-stream kernel hides under the MFMA-bound product kernel of the conv chain when the two run on
different streams -- the situation of the training step, where the pooling of layer l is queued beside the products of layer l + 1.
Reports, per pair: the product kernel alone, the side kernel alone, the two together (both streams busy until both are done) and
the fraction of the shorter one that was hidden:  hidden = (A + B - both) / min(A, B).

  python tools/beside_micro.py [--size 480] [--batch 4] [--grid 24]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wesup_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480)
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--grid', type=int, default=24)
ap.add_argument('--reps', type=int, default=10)
args = ap.parse_args()
d = torch.device('cuda:0')
B, S, g = args.batch, args.size, args.grid
labs = np.stack([synth.voronoi_labels(3 + b, S, S, g) for b in range(B)])
Kmax = (g * g + 63) // 64 * 64
m = ops.sp_preprocess(torch.from_numpy(labs).to(d), None, Kmax)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fa, fb, reps):
    """fa on s1 and fb on s2, reps times each, back to back per stream; the wall time until both streams are idle."""
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    with torch.cuda.stream(s1):
        for _ in range(reps if fa else 0):
            fa()
        ea.record()
    with torch.cuda.stream(s2):
        for _ in range(reps if fb else 0):
            fb()
        eb.record()
    torch.cuda.synchronize()
    return max(e0.elapsed_time(ea), e0.elapsed_time(eb)) / reps * 1e3


def product(K, N, div):
    h = w = S // div
    T = ops.winograd_tiles(B, h, w, 4)
    V = torch.randn(36, T, K, device=d)
    U = torch.randn(36, N, K, device=d) * (1.0 / K) ** 0.5
    y = torch.empty(B, h, w, N, device=d)
    return lambda: ops.winograd_gemm_output_transform(V, U, B, h, w, out=y)


def pool(C, div):
    h = w = S // div
    s = torch.randn(B, h, w, C, device=d)
    out = torch.empty(B, Kmax, C, device=d)
    return lambda: ops.sp_pool_upsample_fwd(s, m, out, 0)


print(f'# B={B} {S}x{S} g={g}; microseconds per pair of launches')
for pname, pa, sname, pb in [('conv1_2 products (K=64)', product(64, 64, 1), 'pool conv1_1', pool(64, 1)),
                             ('conv2_2 products (K=128)', product(128, 128, 2), 'pool conv2_1', pool(128, 2)),
                             ('conv3_2 products (K=256)', product(256, 256, 4), 'pool conv3_1', pool(256, 4))]:
    for f in (pa, pb):
        f()
    a, b_, both = wall(pa, None, args.reps), wall(None, pb, args.reps), wall(pa, pb, args.reps)
    print(f'{pname:26s} {a:7.1f} | {sname:28s} {b_:7.1f} | together {both:7.1f} | hidden {100 * (a + b_ - both) / min(a, b_):5.1f} %')
