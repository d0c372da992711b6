"""Micro driver: superpixel scatter-mean forward/backward at bench shape (B=4, 480x480, 576 SP, 2112 ch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, synth
d = torch.device('cuda:0')
B, H, W, g, C = 4, 480, 480, 24, 2112
mode = sys.argv[1] if len(sys.argv) > 1 else 'voronoi'
mk = synth.skewed_labels if mode == 'skewed' else synth.voronoi_labels
labs = np.stack([mk(b, H, W, g) for b in range(B)])
K = int(labs.max()) + 1
meta = ops.sp_preprocess(torch.from_numpy(labs).to(d), None, K)
fm = torch.randn(B, H, W, C, device=d)
out = torch.empty(B, K, C, device=d)
gup = torch.randn(B, K, C, device=d)
dfm = torch.empty(B, H, W, C, device=d)
for name, fn in (('sp_pool_fwd', lambda: ops.sp_pool_fwd(fm, meta, out=out)), ('sp_pool_bwd', lambda: ops.sp_pool_bwd(gup, meta, out=dfm))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    by = 4.0 * B * (C * H * W + H * W + K * C)
    print(f'{name} [{mode}, K={K}]: {ms*1e3:.1f} us  algorithmic {by/1e9:.3f} GB  {by/ms/1e6:.0f} GB/s')
