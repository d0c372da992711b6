"""Per-layer table of the three conv3x3 kernels at the bench shape (B=4, 480x480): time, TFLOP/s, share of the
ideal (157.3 TFLOP/s) time.  Shows which layers the lost MFMA time sits in."""
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
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
ideal = 0.0
print(f'{"layer":>5} {"HxW":>9} {"ci->co":>9} | {"fwd us":>8} {"TF":>6} | {"dgrad us":>8} {"TF":>6} | {"wgrad us":>8} {"TF":>6} | ideal us')
for l, (ci, co) in enumerate(CONV_CH):
    cin = 4 if l == 0 else ci
    x = torch.randn(B, h, w, cin, device=d)
    wt = torch.randn(co, ci, 3, 3, device=d) * 0.02
    bias = torch.randn(co, device=d)
    dy = torch.randn(B, h, w, co, device=d)
    wf, wd = ops.pack_conv3x3_weight(wt, need_dgrad=(l > 0))
    y = torch.empty(B, h, w, co, device=d)
    dx = torch.empty(B, h, w, cin, device=d)
    dw = torch.empty(co, ci, 3, 3, device=d)
    db = torch.empty(co, device=d)
    fl = 2.0 * B * h * w * ci * co * 9
    t_f = timeit(lambda: ops.conv3x3_fwd(x, wf, bias, co, relu_in=False, out=y))
    t_d = timeit(lambda: ops.conv3x3_dgrad(dy, wd, ci, mask_src=x, out=dx, accumulate=True)) if l > 0 else 0.0
    t_w = timeit(lambda: ops.conv3x3_wgrad(x, dy, ci, relu_in=False, dw=dw, db=db))
    idl = fl / 157.3e12 * 1e6
    ideal += idl
    tot['fwd'] += t_f; tot['dgrad'] += t_d; tot['wgrad'] += t_w
    tf = lambda t: fl / t / 1e9 if t > 0 else 0.0
    print(f'{l:>5} {h:>4}x{w:<4} {ci:>4}->{co:<4} | {t_f*1e3:8.1f} {tf(t_f):6.1f} | {t_d*1e3:8.1f} {tf(t_d):6.1f} | {t_w*1e3:8.1f} {tf(t_w):6.1f} | {idl:7.1f}')
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
    del x, wt, dy, y, dx
print('total ms:', {k: round(v, 3) for k, v in tot.items()}, 'ideal per kernel class %.3f ms' % (ideal / 1e3))
