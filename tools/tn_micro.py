"""Plain TN GEMM (split-K) against the conv wgrad at the same dimensions: separates the TN main loop from the
wgrad's shifted-pixel staging."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
d = torch.device('cuda:0')


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (M, N, K) in [(256, 2304, 57600), (512, 4608, 14400), (128, 1152, 230400), (512, 512, 131072), (1024, 1024, 65536)]:
    A = torch.randn(K, M, device=d); B = torch.randn(K, N, device=d); C = torch.empty(M, N, device=d)
    ms = timeit(lambda: ops.gemm_tn(A, B, out=C))
    print(f'gemm_tn M={M} N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TFLOP/s')
for (B_, H, W, Ci, Co) in [(4, 120, 120, 256, 256), (4, 60, 60, 512, 512), (4, 240, 240, 128, 128)]:
    x = torch.randn(B_, H, W, Ci, device=d); dy = torch.randn(B_, H, W, Co, device=d)
    dw = torch.empty(Co, Ci, 3, 3, device=d); db = torch.empty(Co, device=d)
    ms = timeit(lambda: ops.conv3x3_wgrad(x, dy, Ci, relu_in=True, dw=dw, db=db))
    print(f'wgrad B{B_} {H}x{W} {Ci}->{Co}: {ms*1e3:.1f} us {2.0*B_*H*W*Ci*Co*9/ms/1e9:.1f} TFLOP/s (incl. reduce + bias column sums)')
