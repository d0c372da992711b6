"""Feasibility probe for a Winograd-domain conv forward: the 16 per-position GEMMs (tiles x Cin) . (Cout x Cin)^T have the
tile count and K of ONE plain NT GEMM with M = 16 x tiles; time that at the bench shapes of the deep layers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
for name, T, ci, co, direct_us in (('conv3_2/3', 4 * 60 * 60, 256, 256, 571), ('conv4_1', 4 * 30 * 30, 256, 512, 288),
                                   ('conv4_2/3', 4 * 30 * 30, 512, 512, 567), ('conv5_x', 4 * 15 * 15, 512, 512, 170)):
    M = 16 * T
    A = torch.randn(M, ci, device=d)
    Bw = torch.randn(co, ci, device=d)
    out = torch.empty(M, co, device=d)
    for _ in range(2):
        ops.gemm_nt(A, Bw, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_nt(A, Bw, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * M * ci * co
    print(f'{name:>10} M={M:>7} N={co} K={ci}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF   (direct conv fwd {direct_us} us)')
