"""Time of the NT GEMM against the number of tiles (rounds of 512 block slots) at fixed N, K."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
d = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
tn = (N + 127) // 128
for tiles in (128, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096):
    M = tiles // tn * 128
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
    ops.gemm_nt(A, B, None, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_nt(A, B, None, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'N={N} K={K} tiles={tiles:5d} rounds={tiles/512:5.2f}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s   us/round-equivalent {ms*1e3/(tiles/512):7.1f}')
