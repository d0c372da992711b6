"""Does the NT GEMM slow down because of where the A operand comes from?  Same launch, A rows aliased onto one
row (stride 0: every A fetch hits the cache) against the real matrix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
d = torch.device('cuda:0')


def run(M, N, K, alias):
    A = torch.randn(1, K, device=d).expand(M, K) if alias else torch.randn(M, K, device=d)
    B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
    ops.gemm_nt(A, B, None, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_nt(A, B, None, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'gemm_nt M={M} N={N} K={K} alias={alias}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TFLOP/s')


for (M, N, K) in [(32768, 256, 2304), (8192, 1024, 2304), (32768, 256, 9216), (65536, 128, 2304), (65536, 256, 2304)]:
    for alias in (False, True):
        run(M, N, K, alias)
