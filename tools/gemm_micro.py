"""Micro driver: plain NT GEMM through the C ABI at chosen (M,N,K) - isolates the kernel from conv addressing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
d = torch.device('cuda:0')
shapes = [(32768, 256, k) for k in (288, 576, 1152, 2304, 4608, 9216)] + [(8192, 1024, k) for k in (576, 2304, 9216)] + [(16384, 256, 2304), (16384, 256, 9216)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
    ops.gemm_nt(A, B, None, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_nt(A, B, None, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f'gemm_nt M={M} N={N} K={K} tiles={tiles}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TFLOP/s')
