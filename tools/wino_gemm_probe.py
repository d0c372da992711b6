"""The 16-position batched GEMMs of the Winograd-domain convs at the bench shapes, alone on the GPU: this library's
gemm_nt_kernel<..,3,..> (wesup_gemm_nt_batched) and gemm_tn_kernel<..,3,..> shapes against the vendor's strided-batched
fp32 GEMM (torch.bmm) as a yardstick -- not a product path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from wesup_amd import ops, _lib

d = torch.device('cuda:0')


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


p = lambda t: ctypes.c_void_p(t.data_ptr())
print(f'{"layer":>10} {"tiles":>6} {"Cin":>4} {"Cout":>4} | {"NT ours us":>10} {"TF":>6} | {"bmm us":>8} {"TF":>6}')
for name, T, ci, co in (('conv2_2', 57600, 128, 128), ('conv3_1', 14400, 128, 256), ('conv3_2/3', 14400, 256, 256),
                        ('conv4_1', 3600, 256, 512), ('conv4_2/3', 3600, 512, 512), ('conv5_x', 900, 512, 512)):
    V = torch.randn(16, T, ci, device=d)
    U = torch.randn(16, co, ci, device=d)
    M = torch.empty(16, T, co, device=d)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ours = timeit(lambda: _lib.call('wesup_gemm_nt_batched', p(V), ci, T * ci, p(U), ci, co * ci, p(M), co, T * co, 16, T, co, ci, st))
    Ut = U.transpose(1, 2)
    ref = timeit(lambda: torch.bmm(V, Ut, out=M))
    fl = 2.0 * 16 * T * ci * co
    print(f'{name:>10} {T:>6} {ci:>4} {co:>4} | {ours:10.1f} {fl / ours / 1e6:6.1f} | {ref:8.1f} {fl / ref / 1e6:6.1f}', flush=True)
    del V, U, M
