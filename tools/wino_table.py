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
tot_f, tot_g = [0.0, 0.0], [0.0, 0.0]
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
        line = f'{l:>5} {h:>4}x{w:<4} {ci:>4}->{co:<4} | {t_d*1e3:9.1f} {fl/t_d/1e9:6.1f} | {t_w*1e3:11.1f} {fl/t_w/1e9:8.1f} | {diff:.2e}'
        if ci >= 128:
            wt = torch.randn(co, ci, 3, 3, device=d) * 0.02
            bias = torch.randn(co, device=d)
            wf, wd = ops.pack_conv3x3_weight(wt)
            uf, ud = ops.winograd_pack_weight(wt)
            y0, y1, yr = (torch.empty(B, h, w, co, device=d) for _ in range(3))
            dx0, dx1 = torch.zeros(B, h, w, ci, device=d), torch.zeros(B, h, w, ci, device=d)
            tf0 = timeit(lambda: ops.conv3x3_fwd(x, wf, bias, co, relu_in=False, out=y0, out_relu=yr))
            tf1 = timeit(lambda: ops.conv3x3_fwd_winograd(x, uf, bias, False, out=y1, out_relu=yr))
            td0 = timeit(lambda: ops.conv3x3_dgrad(dy, wd, ci, mask_src=x, out=dx0, accumulate=False))
            td1 = timeit(lambda: ops.conv3x3_dgrad_winograd(dy, ud, mask_src=x, out=dx1, accumulate=False))
            ef = float((y0 - y1).abs().max() / y0.abs().max()); ed = float((dx0 - dx1).abs().max() / dx0.abs().max())
            line += f' | fwd {tf0*1e3:7.1f} -> {tf1*1e3:7.1f} us ({ef:.1e}) | dgrad {td0*1e3:7.1f} -> {td1*1e3:7.1f} us ({ed:.1e})'
            tot_f[0] += tf0; tot_f[1] += min(tf0, tf1); tot_g[0] += td0; tot_g[1] += min(td0, td1)
            del wt, wf, wd, uf, ud, y0, y1, yr, dx0, dx1
        print(line, flush=True)
        del x, dy
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
print('wgrad total ms: direct %.3f, best-of-two per layer %.3f' % (tot_d, tot_w))
print('fwd (layers with Cin >= 128) ms: direct %.3f best %.3f; dgrad: direct %.3f best %.3f' % (*tot_f, *tot_g))
