"""Does a memory-bound kernel overlap with an MFMA-bound one on this chip?  Runs the Winograd-domain batched GEMM of
conv4_2 (stream A) and a Winograd input transform of the same layer (stream B) alone and concurrently, n launches each,
and prints the three wall times; also GEMM || GEMM and transform || transform for reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
B, H, W, C = 4, 60, 60, 512
n = 40
x = torch.randn(B, H, W, C, device=d)
x2 = torch.randn(B, 2 * H, 2 * W, C // 2, device=d)          # a second, larger transform input (conv3_x size)
T = ops.winograd_tiles(B, H, W)
V = torch.randn(16, T, C, device=d)
U = torch.randn(16, C, C, device=d)
M1, M2 = torch.empty(16, T, C, device=d), torch.empty(16, T, C, device=d)
Vo1 = torch.empty(16, ops.winograd_tiles(B, 2 * H, 2 * W), C // 2, device=d)
Vo2 = torch.empty_like(Vo1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


gemm1 = lambda: ops.gemm_nt_batched(V, U, out=M1)
gemm2 = lambda: ops.gemm_nt_batched(V, U, out=M2)
tr1 = lambda: ops.winograd_input_transform(x2, out=Vo1)
tr2 = lambda: ops.winograd_input_transform(x2, out=Vo2)
for f in (gemm1, gemm2, tr1, tr2):
    f()
run(gemm1, tr1)
g = run(gemm1, None); t = run(None, tr1)
print(f'per launch pair, us: GEMM alone {g:.1f}, transform alone {t:.1f}, sum {g + t:.1f}')
print(f'  GEMM || transform  {run(gemm1, tr1):.1f}')
print(f'  GEMM || GEMM       {run(gemm1, gemm2):.1f}   (2 x alone = {2 * g:.1f})')
print(f'  transform || transform {run(tr1, tr2):.1f}   (2 x alone = {2 * t:.1f})')
