"""The memory-bound Winograd transforms of the step alone on the GPU, per layer shape: bytes moved / time against the copy rate.
  python3 tools/transform_micro.py [batch=4] [size=480]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 480


def timed(fn, n=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]


a = torch.empty(256 << 20, dtype=torch.uint8, device=d); b_ = torch.empty_like(a)
t = timed(lambda: b_.copy_(a))
print(f'# batch {B}, {S0} x {S0}; copy of 256 MiB: {2 * a.numel() / t / 1e9:.2f} TB/s (read + write)')
# (layer, divisor of the resolution, channels of the layer input, channels of its output)
layers = [(1, 1, 64, 64), (2, 2, 64, 128), (3, 2, 128, 128), (4, 4, 128, 256), (5, 4, 256, 256), (7, 8, 256, 512), (8, 8, 512, 512), (10, 16, 512, 512)]
tot_in = tot_dual = 0.0
mult = {1: 1, 2: 1, 3: 1, 4: 1, 5: 2, 7: 1, 8: 2, 10: 3}
for l, div, ci, co in layers:
    h = w = S0 // div
    T = ops.winograd_tiles(B, h, w, 4)
    x = torch.randn(B, h, w, ci, device=d); V = torch.empty(36, T, ci, device=d)
    t_in = timed(lambda: ops.winograd_input_transform(x, relu=True, out=V, m=4))
    by_in = 4.0 * (B * h * w * ci + 36 * T * ci)
    dy = torch.randn(B, h, w, co, device=d); dV = torch.empty(36, T, co, device=d); dM = torch.empty(36, T, co, device=d)
    bp = torch.empty(ops.winograd_bias_rows(B, h, w, co), co, device=d)
    t_du = timed(lambda: ops.winograd_dual_transform(dy, dV, dM, bp))
    by_du = 4.0 * (B * h * w * co + 2 * 36 * T * co)
    tot_in += mult[l] * t_in; tot_dual += mult[l] * t_du
    print(f'layer {l:2d} {h:4d}x{w:<4d} {ci:3d}->{co:3d}: input transform {t_in * 1e3:7.1f} us {by_in / t_in / 1e9:5.2f} TB/s   dual transform {t_du * 1e3:7.1f} us {by_du / t_du / 1e9:5.2f} TB/s')
print(f'# per step (12 layers): input transforms {tot_in:.3f} ms, dual transforms {tot_dual:.3f} ms')
