# needs the debug library (the shipped kernels carry no clock / trace code):
#   make -C wesup_amd/csrc debug && WESUP_HIP_LIB=wesup_amd/csrc/libwesup_hip_debug.so python tools/conv_micro.py
"""Micro driver for profiling single GEMM-family kernels (rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops

d = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
B, H, W, Ci, Co = [int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (4, 120, 120, 256, 256))]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
x = torch.randn(B, H, W, Ci, device=d)
w = torch.randn(Co, Ci, 3, 3, device=d) * 0.02
bias = torch.randn(Co, device=d)
dy = torch.randn(B, H, W, Co, device=d)
wf, wd = ops.pack_conv3x3_weight(w)
y = torch.empty(B, H, W, Co, device=d)
dx = torch.empty(B, H, W, Ci, device=d)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(reps + 1):
    if it == 1:
        ev[0].record()
    if which == 'fwd':
        ops.conv3x3_fwd(x, wf, bias, Co, False, out=y)
    elif which == 'dgrad':
        ops.conv3x3_dgrad(dy, wd, Ci, mask_src=x, out=dx, accumulate=True)
    else:
        ops.conv3x3_wgrad(x, dy, Ci, False)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(f'{which} B{B} {H}x{W} {Ci}->{Co}: {ms*1e3:.1f} us  {2.0*B*H*W*Ci*Co*9/ms/1e9:.1f} TFLOP/s')

import ctypes
from wesup_amd import _lib
mhz = ctypes.c_double(0.0)
_lib.load().wesup_debug_clock(ctypes.byref(mhz))
if which != 'wgrad':
    print(f'  in-kernel clock of the last launch: {mhz.value:.0f} MHz -> fp32 MFMA peak at that clock {157.3 * mhz.value / 2400:.1f} TFLOP/s')
