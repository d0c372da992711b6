"""Implicit GEMM vs Winograd F(2x2,3x3) vs F(4x4,3x3) per VGG layer and pass (forward, input gradient, weight gradient), each
alone on the GPU: microseconds and the best algorithm -- the measured table behind wesup_amd.engine.default_route.

  python tools/wino_table.py [--size 480] [--batch 4] [--reps 5]      # profiles/r03_wino_table_<size>.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480)
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--min-ci', type=int, default=64, help='layers with fewer input channels are listed for the direct form only')
args = ap.parse_args()
d = torch.device('cuda:0')
B, H, W, reps = args.batch, args.size, args.size, args.reps


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


h, w = H, W
tot = {k: [0.0, 0.0, 0.0, 0.0] for k in ('fwd', 'dgrad', 'wgrad')}       # direct, F2, F4, best
print(f'# B={B} {H}x{W}; us per launch alone on the GPU (direct / F(2x2) / F(4x4)); error of the Winograd forms against the '
      'direct kernel, fraction of the tensor max')
print(f'{"layer":>5} {"HxW":>9} {"ci->co":>9} | {"fwd":^26} | {"dgrad":^26} | {"wgrad":^26} | best m (fwd dgrad wgrad) | err F2 / F4 (fwd, wgrad)')
for l, (ci, co) in enumerate(CONV_CH):
    if l > 0:
        x = torch.relu(torch.randn(B, h, w, ci, device=d))
        dy = torch.randn(B, h, w, co, device=d)
        wt = torch.randn(co, ci, 3, 3, device=d) * (2.0 / (9 * ci)) ** 0.5
        bias = torch.randn(co, device=d)
        wf, wd = ops.pack_conv3x3_weight(wt)
        y = [torch.empty(B, h, w, co, device=d) for _ in range(3)]
        yr = torch.empty(B, h, w, co, device=d)
        dx = [torch.zeros(B, h, w, ci, device=d) for _ in range(3)]
        dw = [torch.empty(co, ci, 3, 3, device=d) for _ in range(3)]
        db = torch.empty(co, device=d)
        t = {'fwd': [0.0] * 3, 'dgrad': [0.0] * 3, 'wgrad': [0.0] * 3}
        t['fwd'][0] = timeit(lambda: ops.conv3x3_fwd(x, wf, bias, co, relu_in=False, out=y[0], out_relu=yr))
        t['dgrad'][0] = timeit(lambda: ops.conv3x3_dgrad(dy, wd, ci, mask_src=x, out=dx[0], accumulate=False))
        t['wgrad'][0] = timeit(lambda: ops.conv3x3_wgrad(x, dy, ci, relu_in=False, dw=dw[0], db=db))
        errs = ['-', '-']
        if ci >= args.min_ci:
            for k, m in ((1, 2), (2, 4)):
                uf, ud = ops.winograd_pack_weight(wt, m=m)
                v = torch.empty(ops.winograd_positions(m), ops.winograd_tiles(B, h, w, m), ci, device=d)
                t['fwd'][k] = timeit(lambda: ops.conv3x3_fwd_winograd(x, uf, bias, False, out=y[k], v_keep=v, m=m))
                t['dgrad'][k] = timeit(lambda: ops.conv3x3_dgrad_winograd(dy, ud, mask_src=x, out=dx[k], accumulate=False, m=m))
                t['wgrad'][k] = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, relu_in=False, dw=dw[k], db=db, v_pre=v, m=m))
                errs[k - 1] = f'{rel(y[k], y[0]):.1e},{rel(dw[k], dw[0]):.1e}'
                del uf, ud, v
        best = []
        cells = []
        for p in ('fwd', 'dgrad', 'wgrad'):
            cand = [(v, i) for i, v in enumerate(t[p]) if v > 0]
            bv, bi = min(cand)
            best.append((0, 2, 4)[bi])
            for i in range(3):
                tot[p][i] += t[p][i] if t[p][i] > 0 else t[p][0]
            tot[p][3] += bv
            cells.append(' '.join(f'{v:8.1f}' if v > 0 else f'{"-":>8}' for v in t[p]))
        print(f'{l:>5} {h:>4}x{w:<4} {ci:>4}->{co:<4} | ' + ' | '.join(cells) + f' | {best[0]} {best[1]} {best[2]} | {errs[0]} / {errs[1]}',
              flush=True)
        del x, dy, wt, wf, wd, y, yr, dx, dw
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
for p in ('fwd', 'dgrad', 'wgrad'):
    print(f'{p:>6} total ms (layers 1..12): direct {tot[p][0]/1e3:.3f}  F(2x2) where available {tot[p][1]/1e3:.3f}  '
          f'F(4x4) where available {tot[p][2]/1e3:.3f}  best per layer {tot[p][3]/1e3:.3f}')
