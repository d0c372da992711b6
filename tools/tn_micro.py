"""The TN products of the step's weight gradients alone on the GPU, per layer shape (operands exist: the forward's V, the dual
transform's dM): ops.conv3x3_wgrad_winograd_pre = batched TN over the 36 positions + wino_wgrad_reduce; plus plain TN GEMMs of
growing size (what the kernel reaches when quantisation does not matter).   python3 tools/tn_micro.py [batch=4] [size=480]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 480


def timed(fn, n=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]


print(f'# batch {B}, {S0} x {S0}; WESUP_TN_S1_MIN_TILES={os.environ.get("WESUP_TN_S1_MIN_TILES")}')
layers = [(1, 1, 64, 64), (2, 2, 64, 128), (3, 2, 128, 128), (4, 4, 128, 256), (5, 4, 256, 256), (7, 8, 256, 512), (8, 8, 512, 512), (10, 16, 512, 512)]
for l, div, ci, co in layers:
    h = w = S0 // div
    T = ops.winograd_tiles(B, h, w, 4)
    V = torch.randn(36, T, ci, device=d); dM = torch.randn(36, T, co, device=d)
    rows = ops.winograd_bias_rows(B, h, w, co)
    bp = torch.randn(rows, co, device=d)
    dw = torch.empty(co, ci, 3, 3, device=d); db = torch.empty(co, device=d)
    t = timed(lambda: ops.conv3x3_wgrad_winograd_pre(V, dM, bp, B, h, w, dw, db))
    fl = 2.0 * 36 * T * ci * co
    by = 4.0 * 36 * T * (ci + co)
    print(f'layer {l:2d} {h:4d}x{w:<4d} {ci:3d}->{co:3d} tiles {T:6d}: {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TF  operands {by / t / 1e9:6.2f} TB/s')
for M, N, K in [(1024, 1024, 2560), (1024, 2112, 2560), (2048, 2048, 8192), (4096, 4096, 8192), (256, 512, 14400), (3600, 768, 640)]:
    A = torch.randn(K, M, device=d); Bm = torch.randn(K, N, device=d); out = torch.empty(M, N, device=d)
    t = timed(lambda: ops.gemm_tn(A, Bm, out=out))
    print(f'plain TN {M} x {N} x {K}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:6.1f} TF')
