"""The epilogue forms of the one-kernel Winograd product route (csrc/wino_fused.hip) alone, on the two shapes whose input
gradients end the backward pass: us per launch.   python tools/epilogue_micro.py [--batch 4] [--size 480]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops, _lib
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=4); ap.add_argument('--size', type=int, default=480)
ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
d = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3
B, S = a.batch, a.size
Kmax = 576
for name, K, N, div in (('conv1_2 (64 -> 64, %d^2)' % S, 64, 64, 1), ('conv2_1 dgrad (128 -> 64, %d^2, unpooling to %d^2)' % (S // 2, S), 128, 64, 2)):
    H = W = S // div
    T = ops.winograd_tiles(B, H, W, 4)
    V = torch.randn(36, T, K, device=d); U = torch.randn(36, N, K, device=d) * K ** -0.5
    bias = torch.randn(N, device=d)
    y = torch.empty(B, H, W, N, device=d)
    bits = torch.randint(0, 16, (B, H, W, N // 4), dtype=torch.uint8, device=d)
    side = torch.randn(B, Kmax, N, device=d)
    Hu, Wu = 2 * H, 2 * W
    def call(bias=None, mask_bits=None, y=None, y_pool=None, pool_code=None, up_code=None, up_dst=None, gather=None, accumulate=0, hu=0, wu=0):
        hw = (hu * wu) if up_dst is not None else H * W
        row = area = None
        if gather is not None:
            row, area = gather
        _lib.call('wesup_winograd_gemm_output_transform_ex', P(V), 0, P(U), P(bias), None, P(mask_bits), P(y), P(y_pool), 1 if y_pool is not None else 0,
                  P(pool_code), None, P(up_code), P(up_dst), hu, wu, P(side if gather is not None else None), P(row), P(area), Kmax if gather is not None else 0,
                  B, H, W, K, N, accumulate, st)
    print(name, f'tiles {T}')
    print(f'  plain (bias)                         {timeit(lambda: call(bias=bias, y=y)):7.1f} us')
    if div == 1:
        yp = torch.empty(B, H // 2, W // 2, N, device=d); pc = torch.empty(B, H // 2, W // 2, N // 4, dtype=torch.int16, device=d)
        print(f'  forward: + pooled output + codes     {timeit(lambda: call(bias=bias, y=y, y_pool=yp, pool_code=pc)):7.1f} us')
        print(f'  dgrad: mask bits + accumulate        {timeit(lambda: call(mask_bits=bits, y=y, accumulate=1)):7.1f} us')
        row = torch.randint(0, Kmax, (B, H, W), dtype=torch.int32, device=d); area = torch.randint(100, 500, (B, Kmax), dtype=torch.int32, device=d)
        # (a Voronoi-like map: neighbouring pixels share a row)
        row = (torch.arange(H, device=d)[:, None] // 20 * 24 + torch.arange(W, device=d)[None, :] // 20).to(torch.int32).expand(B, H, W).contiguous()
        print(f'  dgrad: mask bits + gathered side grad{timeit(lambda: call(mask_bits=bits, y=y, gather=(row, area))):7.1f} us')
        print(f'  ... rows scaled beforehand           {timeit(lambda: call(mask_bits=bits, y=y, gather=(row, None))):7.1f} us')
    else:
        dst = torch.zeros(B, Hu, Wu, N, device=d); code = torch.randint(0, 4 ** 4, (B, H, W, N // 4), dtype=torch.int16, device=d)
        print(f'  dgrad: unpool by codes               {timeit(lambda: call(up_code=code, up_dst=dst, hu=Hu, wu=Wu)):7.1f} us')
        row = (torch.arange(Hu, device=d)[:, None] // 20 * 24 + torch.arange(Wu, device=d)[None, :] // 20).to(torch.int32).expand(B, Hu, Wu).contiguous()
        area = torch.randint(100, 500, (B, Kmax), dtype=torch.int32, device=d)
        print(f'  dgrad: unpool by codes + gather      {timeit(lambda: call(up_code=code, up_dst=dst, hu=Hu, wu=Wu, gather=(row, area))):7.1f} us')
        print(f'  ... rows scaled beforehand           {timeit(lambda: call(up_code=code, up_dst=dst, hu=Hu, wu=Wu, gather=(row, None))):7.1f} us')
