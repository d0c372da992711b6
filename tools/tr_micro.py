import os, sys
sys.path.insert(0, os.getcwd())
import torch
from wesup_amd import ops
d = torch.device('cuda:0')
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('layout', 'blocked' if os.environ.get('WESUP_VBLOCKED') else 'planes')
for (h, c) in ((480, 64), (240, 64), (240, 128), (120, 128), (120, 256), (60, 256), (60, 512), (30, 512)):
    B = 4
    x = torch.randn(B, h, h, c, device=d)
    T = ops.winograd_tiles(B, h, h, 4)
    buf = torch.empty(36 * ((T + 31) // 32 * 32) * c, device=d)      # (room for the blocked layout's last, partial tile block)
    V = buf[:36 * T * c].view(36, T, c)
    us = timeit(lambda: ops.winograd_input_transform(x, relu=True, out=V, m=4))
    by = 4.0 * (B * h * h + 36 * T) * c
    print(f'{h:4d}^2 x {c:3d}: {us:7.1f} us  {by / us * 1e-6:5.2f} TB/s')
