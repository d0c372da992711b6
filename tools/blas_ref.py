"""What the vendor GEMM (torch.matmul fp32 -> hipBLASLt / rocBLAS) reaches on the conv-equivalent shapes: a yardstick
for the hand-written kernels, not part of the product path."""
import torch
d = torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (M, N, K) in [(57600, 256, 2304), (14400, 512, 4608), (230400, 128, 1152), (921600, 64, 576), (3600, 512, 4608), (32768, 256, 2304), (65536, 256, 2304), (8192, 1024, 2304), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d)
    ms = t(lambda: torch.matmul(A, B.t()))
    print(f'M={M} N={N} K={K}: NT {2.0*M*N*K/ms/1e9:.1f} TF')
    del A, B
for (P, Co, Ci9) in [(57600, 256, 2304), (14400, 512, 4608), (230400, 128, 1152)]:
    dy = torch.randn(P, Co, device=d); x = torch.randn(P, Ci9, device=d)
    ms = t(lambda: torch.matmul(dy.t(), x))
    print(f'TN pixels={P} Co={Co} 9Ci={Ci9}: {2.0*P*Co*Ci9/ms/1e9:.1f} TF')
