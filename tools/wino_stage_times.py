"""The three passes of an F(4x4,3x3)-domain conv, each alone on the GPU: us and GB/s of its algorithmic bytes.
  python tools/wino_stage_times.py [--size 480] [--batch 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480); ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--layers', default='1,2,3,4,5,7,8,10')
a = ap.parse_args()
d = torch.device('cuda:0')
B, H, W, m = a.batch, a.size, a.size, 4


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


h, w = H, W
want = {int(v) for v in a.layers.split(',')}
print(f'{"layer":>5} {"HxW":>9} {"ci->co":>9} | in-transform us (GB/s) | gemm us (TF, GB/s) | out-transform us (GB/s) | sum')
for l, (ci, co) in enumerate(CONV_CH):
    if l in want:
        x = torch.relu(torch.randn(B, h, w, ci, device=d))
        T, P = ops.winograd_tiles(B, h, w, m), 36
        V = torch.empty(P, T, ci, device=d); Mt = torch.empty(P, T, co, device=d)
        u = torch.randn(P, co, ci, device=d) * 0.05
        y = torch.empty(B, h, w, co, device=d)
        bias = torch.randn(co, device=d)
        t_in = timeit(lambda: ops.winograd_input_transform(x, out=V, m=m))
        t_g = timeit(lambda: ops.gemm_nt_batched(V, u, out=Mt))
        t_out = timeit(lambda: ops.winograd_output_transform(Mt, B, h, w, bias=bias, out=y, m=m))
        b_in = 4.0 * (x.numel() + V.numel()); b_g = 4.0 * (V.numel() + Mt.numel() + u.numel()); b_out = 4.0 * (Mt.numel() + y.numel())
        fl = 2.0 * P * T * ci * co
        print(f'{l:>5} {h:>4}x{w:<4} {ci:>4}->{co:<4} | {t_in:7.1f} ({b_in / t_in / 1e3:6.0f}) | {t_g:7.1f} ({fl / t_g / 1e6:5.1f}, {b_g / t_g / 1e3:6.0f}) | '
              f'{t_out:7.1f} ({b_out / t_out / 1e3:6.0f}) | {t_in + t_g + t_out:7.1f}', flush=True)
        del x, V, Mt, u, y
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
