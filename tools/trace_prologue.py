# needs the debug library (the shipped kernels carry no clock / trace code):
#   make -C wesup_amd/csrc debug && WESUP_HIP_LIB=wesup_amd/csrc/libwesup_hip_debug.so python tools/trace_prologue.py
import sys, os, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from wesup_amd import ops, _lib
d = torch.device('cuda:0'); lib = _lib.load()
def run(name, fn, nblocks):
    buf = torch.zeros(nblocks * 6, dtype=torch.int64, device=d)
    fn(); torch.cuda.synchronize()
    lib.wesup_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.wesup_debug_set_trace(None)
    raw = buf.cpu().numpy().reshape(nblocks, 6).astype(np.float64) * 0.01
    st, ls, le, en, pre, iss = (raw[:, i] for i in range(6))
    print(f'{name}: setup (start->before DMA issue) p50 {np.median(pre-st):.2f} p90 {np.percentile(pre-st,90):.2f} | issue p50 {np.median(iss-pre):.2f} | wait+barrier p50 {np.median(ls-iss):.2f} p90 {np.percentile(ls-iss,90):.2f} | loop p50 {np.median(le-ls):.1f} | epilogue p50 {np.median(en-le):.2f}')
x = torch.randn(4, 480, 480, 64, device=d); w = torch.randn(64, 64, 3, 3, device=d) * 0.05
wf, wd = ops.pack_conv3x3_weight(w); y = torch.empty(4, 480, 480, 64, device=d); bias = torch.zeros(64, device=d)
run('conv1_2 fwd', lambda: ops.conv3x3_fwd(x, wf, bias, 64, True, out=y), 7200)
x = torch.randn(4, 120, 120, 256, device=d); w = torch.randn(256, 256, 3, 3, device=d) * 0.02
wf, _ = ops.pack_conv3x3_weight(w, need_dgrad=False); y = torch.empty(4, 120, 120, 256, device=d); bias = torch.zeros(256, device=d)
run('conv L6 fwd', lambda: ops.conv3x3_fwd(x, wf, bias, 256, True, out=y), 512)
M, N, K = 32768, 256, 2304
A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
run('gemm', lambda: ops.gemm_nt(A, B, None, out=C), 512)
M, N, K = 921600, 64, 576
A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
run('gemm N64 K576 (7200 tiles)', lambda: ops.gemm_nt(A, B, None, out=C), 7200)
