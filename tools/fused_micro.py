"""The one-kernel Winograd product route (csrc/wino_fused.hip) alone on the GPU, on the product shapes of the training step:
microseconds per launch, executed TFLOP/s, bytes past the kernel's own model, and the distance from the two-kernel route.

  python tools/fused_micro.py [--size 480] [--batch 4] [--reps 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480)
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--check', action='store_true', help='compare with the batched GEMM + output transform')
ap.add_argument('--unfused', action='store_true', help='also time the two-kernel route (batched GEMM + output transform) on the same operands')
ap.add_argument('--only', default='', help='substring of the shape names to run (counter passes)')
args = ap.parse_args()
d = torch.device('cuda:0')
B, S = args.batch, args.size
# (name, K, N, resolution divisor): the products of a step that take this route
SHAPES = [('conv1_2 fwd/dgrad', 64, 64, 1), ('conv2_1 fwd', 64, 128, 2), ('conv2_1 dgrad', 128, 64, 2), ('conv2_2 fwd/dgrad', 128, 128, 2),
          ('conv3_1 fwd', 128, 256, 4), ('conv3_1 dgrad', 256, 128, 4), ('conv3_2 fwd/dgrad', 256, 256, 4), ('conv4_1 fwd', 256, 512, 8)]
COUNT = {'conv1_2 fwd/dgrad': 2, 'conv2_1 fwd': 1, 'conv2_1 dgrad': 1, 'conv2_2 fwd/dgrad': 2, 'conv3_1 fwd': 1, 'conv3_1 dgrad': 1,
         'conv3_2 fwd/dgrad': 4, 'conv4_1 fwd': 1}


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'# B={B} {S}x{S}')
total = 0.0
for name, K, N, div in SHAPES:
    if args.only and args.only not in name:
        continue
    h = w = S // div
    T = ops.winograd_tiles(B, h, w, 4)
    V = torch.randn(36, T, K, device=d)
    U = torch.randn(36, N, K, device=d) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device=d)
    y = torch.empty(B, h, w, N, device=d)
    us = timeit(lambda: ops.winograd_gemm_output_transform(V, U, B, h, w, bias=bias, out=y), args.reps)
    gf = 2.0 * 36 * T * K * N * 1e-9
    mb = (36 * T * K + B * h * w * N + 36 * N * K) * 4e-6
    err = ''
    if args.check:
        Mt = ops.gemm_nt_batched(V, U)
        ref = ops.winograd_output_transform(Mt, B, h, w, bias=bias, m=4)
        err = f'  err {float((y - ref).abs().max() / ref.abs().max()):.1e}'
        # ragged tile count: a second, odd-sized problem through the same instantiation
        h2, w2 = h - 3, w - 5
        T2 = ops.winograd_tiles(1, h2, w2, 4)
        V2 = torch.randn(36, T2, K, device=d)
        y2 = ops.winograd_gemm_output_transform(V2, U, 1, h2, w2, bias=bias)
        ref2 = ops.winograd_output_transform(ops.gemm_nt_batched(V2, U), 1, h2, w2, bias=bias, m=4)
        err += f' / {float((y2 - ref2).abs().max() / ref2.abs().max()):.1e}'
    if args.unfused:
        Mt = torch.empty(36, T, N, device=d)
        def two():
            ops.gemm_nt_batched(V, U, out=Mt)
            ops.winograd_output_transform(Mt, B, h, w, bias=bias, out=y, m=4)
        us2 = timeit(two, args.reps)
        err += f'   two-kernel route {us2:7.1f} us  ({((T + 31) // 32) * (N // 64)} blocks of the one-kernel route)'
    total += us * COUNT[name]
    print(f'{name:18s} K={K:3d} N={N:3d} {h:3d}x{w:<3d} tiles {T:6d}: {us:7.1f} us  {gf / us * 1e3:6.1f} TF  {mb / us:5.2f} TB/s{err}')
print(f'sum over the 13 launches of a step: {total * 1e-3:.3f} ms')
