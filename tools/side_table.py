"""Per-layer table of the side-branch GEMMs at the bench shape (B=4, 480x480): the 1x1 side conv forward
(gemm_nt, M = pixels, K = C, N = C/2), its input gradient (gemm_nt, K = C/2, N = C, written into G_l) and its weight
gradient (gemm_tn, K = pixels).  Memory bound at the shallow layers: the ideal column is bytes / 5 TB/s or
FLOPs / 125 TFLOP/s, whichever is larger."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER

d = torch.device('cuda:0')
B, H, W = 4, 480, 480
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


h, w = H, W
tot = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
print(f'{"layer":>5} {"pixels":>8} {"C":>4} | {"fwd us":>7} {"ideal":>6} | {"dgrad us":>8} {"ideal":>6} | {"wgrad us":>8} {"ideal":>6}')
for l, (ci, co) in enumerate(CONV_CH):
    P = B * h * w
    y = torch.randn(P, co, device=d)
    ws = torch.randn(co // 2, co, device=d) * 0.05
    wsT = ws.t().contiguous()
    bias = torch.randn(co // 2, device=d)
    s = torch.empty(P, co // 2, device=d)
    ds = torch.randn(P, co // 2, device=d)
    G = torch.empty(P, co, device=d)
    dws = torch.empty(co // 2, co, device=d)
    dbs = torch.empty(co // 2, device=d)
    fl = 2.0 * P * co * (co // 2)
    t_f = timeit(lambda: ops.gemm_nt(y, ws, bias, out=s))
    t_d = timeit(lambda: ops.gemm_nt(ds, wsT, None, out=G))
    t_w = timeit(lambda: ops.gemm_tn(ds, y, out=dws, ws_tag='side', colsum=dbs))
    by_f = (P * co + P * co // 2) * 4
    by_d = (P * co // 2 + P * co) * 4
    by_w = (P * co // 2 + P * co) * 4
    ideal = lambda by: max(by / 5e12, fl / 125e12) * 1e6
    for i, v in enumerate((t_f, ideal(by_f), t_d, ideal(by_d), t_w, ideal(by_w))):
        tot[i] += v
    print(f'{l:>5} {P:>8} {co:>4} | {t_f:7.1f} {ideal(by_f):6.1f} | {t_d:8.1f} {ideal(by_d):6.1f} | {t_w:8.1f} {ideal(by_w):6.1f}')
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
    del y, s, ds, G
print('total us: fwd %.0f (ideal %.0f)  dgrad %.0f (ideal %.0f)  wgrad %.0f (ideal %.0f)' % tuple(tot))
