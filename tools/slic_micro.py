"""Timing of the GPU SLIC at bench shape (B=4, 480x480, n_segments = HW/200)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, synth
d = torch.device('cuda:0')
B, H, W = 4, 480, 480
img = torch.from_numpy(np.stack([synth.synth_image(b, H, W) for b in range(B)])).to(d)
n_seg = H * W // 200
lab, n = ops.slic(img, n_seg); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): lab, n = ops.slic(img, n_seg)
e1.record(); torch.cuda.synchronize()
print(f'wesup_slic B={B} {H}x{W} n_segments={n_seg}: {e0.elapsed_time(e1)/10:.3f} ms per batch; superpixels per image {n.tolist()}')
