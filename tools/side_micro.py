"""Micro driver: the memory-bound kernels of the commuted side branch at the bench shape, each alone on the GPU:
fused upsample + superpixel mean of a conv output (forward) and its backward into the conv's gradient buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, synth
d = torch.device('cuda:0')
B, H, W, g = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 480, 480, 24)))
labs = np.stack([synth.voronoi_labels(b, H, W, g) for b in range(B)])
K = int(labs.max()) + 1
meta = ops.sp_preprocess(torch.from_numpy(labs).to(d), None, K)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


junk = torch.empty(600 << 20, device=d)     # flush the caches between layers
for (s, C) in ((1, 64), (2, 128), (4, 256)):
    h, w = H // s, W // s
    y = torch.randn(B, h, w, C, device=d)
    out = torch.empty(B, K, C, device=d)
    gb = torch.randn(B, K, C, device=d)
    G = torch.empty(B, h, w, C, device=d)
    by = 4.0 * B * (h * w * C + H * W + K * C)
    junk.zero_()
    ms = timeit(lambda: ops.sp_pool_upsample_fwd(y, meta, out, 0))
    print(f'{h}x{w} C={C}: pool fwd {ms * 1e3:7.1f} us {by / ms / 1e6:7.0f} GB/s', end='   ')
    ms = timeit(lambda: ops.upsample_bwd_fused(gb, meta.new_row, meta.area_new, H, W, 0, h, w, C, out=G))
    print(f'bwd {ms * 1e3:7.1f} us {by / ms / 1e6:7.0f} GB/s', end='')
    if s > 1:
        n = 2 if s == 2 else 3
        gs = [torch.randn(B, K, C, device=d) for _ in range(n)]
        Gs = [torch.empty(B, h, w, C, device=d) for _ in range(n)]
        ms = timeit(lambda: ops.upsample_bwd_fused_group(gs, meta.new_row, meta.area_new, H, W, h, w, Gs))
        print(f'   bwd x{n} in one launch {ms * 1e3:7.1f} us {n * by / ms / 1e6:7.0f} GB/s', end='')
    print()
