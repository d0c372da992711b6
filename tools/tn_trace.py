# needs the debug library (the shipped kernels carry no clock / trace code):
#   make -C wesup_amd/csrc debug && WESUP_HIP_LIB=wesup_amd/csrc/libwesup_hip_debug.so python tools/tn_trace.py
"""Per-block timeline of one conv3x3 wgrad launch (debug trace in the TN kernel): start order, which blocks share a
CU, when each ends."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, _lib
d = torch.device('cuda:0')
lib = _lib.load()
B, H, W, Ci, Co = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (4, 120, 120, 256, 256))]
x = torch.randn(B, H, W, Ci, device=d); dy = torch.randn(B, H, W, Co, device=d)
dw = torch.empty(Co, Ci, 3, 3, device=d); db = torch.empty(Co, device=d)
fn = lambda: ops.conv3x3_wgrad(x, dy, Ci, relu_in=True, dw=dw, db=db)
nmax = 4096
buf = torch.zeros(nmax * 6, dtype=torch.int64, device=d)
fn(); torch.cuda.synchronize()
lib.wesup_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.wesup_debug_set_trace(None)
raw = buf.cpu().numpy().reshape(nmax, 6)
n = int((raw[:, 0] != 0).sum())
raw = raw[:n]
t = raw[:, :4].astype(np.float64) * 0.01
xcc = raw[:, 4] & 0xf; hw = raw[:, 5]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
t0 = t[:, 0].min()
st, ls, le, en = (t[:, i] - t0 for i in range(4))
print(f'wgrad B{B} {H}x{W} {Ci}->{Co}: {n} blocks, span {en.max():.1f} us')
print(f'  start p50 {np.median(st):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f};  prologue p50 {np.median(ls-st):.1f};  epilogue p50 {np.median(en-le):.1f} max {(en-le).max():.1f}')
print(f'  loop  p10 {np.percentile(le-ls,10):.1f} p50 {np.median(le-ls):.1f} p90 {np.percentile(le-ls,90):.1f} max {(le-ls).max():.1f}')
print(f'  end   p10 {np.percentile(en,10):.1f} p50 {np.median(en):.1f} p90 {np.percentile(en,90):.1f} max {en.max():.1f}')
key = xcc * 1000 + se * 100 + sh * 16 + cu
groups = {}
for i, k_ in enumerate(key.tolist()): groups.setdefault(k_, []).append(i)
print(f'  distinct CUs {len(groups)}, blocks per CU histogram {np.bincount([len(v) for v in groups.values()]).tolist()}')
pairs = [v for v in groups.values() if len(v) == 2]
first_end = np.array([min(en[v[0]], en[v[1]]) for v in pairs]); last_end = np.array([max(en[v[0]], en[v[1]]) for v in pairs])
older_first = np.mean([en[min(v, key=lambda i: st[i])] < en[max(v, key=lambda i: st[i])] for v in pairs])
print(f'  same-CU pairs {len(pairs)}: first of a pair ends p50 {np.median(first_end):.1f}, second p50 {np.median(last_end):.1f}; the earlier-started block ends first in {100*older_first:.0f} %')
idd = np.array([[min(v), max(v)] for v in pairs])
print(f'  linear block ids of a pair differ by: p10 {np.percentile(idd[:,1]-idd[:,0],10):.0f} p50 {np.median(idd[:,1]-idd[:,0]):.0f} p90 {np.percentile(idd[:,1]-idd[:,0],90):.0f}')
lo = np.arange(n) < n // 2
print(f'  blocks with id < n/2: end p50 {np.median(en[lo]):.1f};  id >= n/2: end p50 {np.median(en[~lo]):.1f}')
