"""conv3x3 forward with and without ReLU-on-load, against the plain GEMM of the same shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
d = torch.device('cuda:0')
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (B, H, W, C) in [(4, 128, 64, 256), (8, 128, 128, 256), (4, 120, 120, 256)]:
    x = torch.randn(B, H, W, C, device=d); w = torch.randn(C, C, 3, 3, device=d) * 0.02
    wf, wd = ops.pack_conv3x3_weight(w); y = torch.empty(B, H, W, C, device=d); bias = torch.randn(C, device=d)
    M, K = B * H * W, 9 * C
    A = torch.randn(M, K, device=d); Bw = torch.randn(C, K, device=d); Cm = torch.empty(M, C, device=d)
    fl = 2.0 * M * C * K
    r = [fl / t(lambda: ops.conv3x3_fwd(x, wf, bias, C, rl, out=y)) / 1e9 for rl in (False, True)]
    g = fl / t(lambda: ops.gemm_nt(A, Bw, bias, out=Cm)) / 1e9
    g2 = fl / t(lambda: ops.gemm_nt(A, Bw, bias, out=Cm, flags=ops.RELU_IN)) / 1e9
    print(f'B{B} {H}x{W} C{C} ({M//128*C//128} tiles): conv no-relu {r[0]:.1f} TF, conv relu-in {r[1]:.1f} TF | plain GEMM {g:.1f} TF, plain GEMM relu-in {g2:.1f} TF')
