"""Direct vs Winograd-domain conv3x3 weight gradient per layer at the bench shape (B=4, 480x480), each alone on the
GPU: time and the direct-form TFLOP/s equivalent.  Decides which layers the engine routes through
wesup_conv3x3_wgrad_winograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER

d = torch.device('cuda:0')
B, H, W = 4, 480, 480
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


h, w = H, W
tot_d = tot_w = 0.0
print(f'{"layer":>5} {"HxW":>9} {"ci->co":>9} | {"direct us":>9} {"TF":>6} | {"winograd us":>11} {"TF-equiv":>8} | max rel diff')
for l, (ci, co) in enumerate(CONV_CH):
    if l > 0:
        x = torch.relu(torch.randn(B, h, w, ci, device=d))
        dy = torch.randn(B, h, w, co, device=d)
        dw0 = torch.empty(co, ci, 3, 3, device=d)
        dw1 = torch.empty(co, ci, 3, 3, device=d)
        db = torch.empty(co, device=d)
        fl = 2.0 * B * h * w * ci * co * 9
        t_d = timeit(lambda: ops.conv3x3_wgrad(x, dy, ci, relu_in=False, dw=dw0, db=db))
        t_w = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, relu_in=False, dw=dw1, db=db))
        diff = float((dw0 - dw1).abs().max() / dw0.abs().max())
        tot_d += t_d; tot_w += min(t_d, t_w)
        print(f'{l:>5} {h:>4}x{w:<4} {ci:>4}->{co:<4} | {t_d*1e3:9.1f} {fl/t_d/1e9:6.1f} | {t_w*1e3:11.1f} {fl/t_w/1e9:8.1f} | {diff:.2e}')
        del x, dy
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
print('total ms: direct %.3f, best-of-two per layer %.3f' % (tot_d, tot_w))
