"""wesup_gemm_nt_batched alone on the shapes of the step's K = 512 Winograd products (WESUP_NTB_SHAPE=0/1/2 picks the tile rule).
  WESUP_NTB_SHAPE=0 python3 tools/ntb_micro.py ; WESUP_NTB_SHAPE=2 python3 tools/ntb_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
print('# WESUP_NTB_SHAPE =', os.environ.get('WESUP_NTB_SHAPE', '(default rule)'))
for nb, M, N, K in [(36, 900, 512, 512), (36, 900, 256, 512), (36, 256, 512, 512), (36, 2500, 512, 512), (36, 8192, 512, 512)]:
    A = torch.randn(nb, M, K, device=d); Bw = torch.randn(nb, N, K, device=d); out = torch.empty(nb, M, N, device=d)
    for _ in range(5):
        ops.gemm_nt_batched(A, Bw, out)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nt_batched(A, Bw, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = sorted(ts)[len(ts) // 2]
    print(f'{nb} x ({M} x {N} x {K}): {t * 1e3:8.1f} us  {2.0 * nb * M * N * K / t / 1e9:7.1f} TF')
