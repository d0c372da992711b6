"""When does a stream that waits for an event of another stream resume -- at the recorded point, or later?
Stream A: a1 (short), RECORD E, a2 (long).  Stream B: WAIT E, b1.  Timing events bracket a1's end and b1's start.
  python tools/wait_probe.py
"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops, _lib
dev = torch.device('cuda:0')
A, Bs, Cs = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev)
big = torch.randn(16384, 8192, device=dev)
small = torch.randn(256, 256, device=dev)

def busy_long():
    return big @ big.t()[:, :4096]          # ~1 ms
def busy_short():
    return small @ small

def trial(kind, third_busy):
    torch.cuda.synchronize()
    t_a1 = torch.cuda.Event(enable_timing=True); t_b1 = torch.cuda.Event(enable_timing=True); t_a2 = torch.cuda.Event(enable_timing=True)
    if third_busy:
        with torch.cuda.stream(Cs):
            for _ in range(3): busy_long()
    with torch.cuda.stream(A):
        busy_short()
        if kind == 'torch':
            e = torch.cuda.Event(); e.record()
        elif kind == 'torch_timing':
            e = torch.cuda.Event(enable_timing=True); e.record()
        else:
            ops.sync_record(200)
        t_a1.record()
        busy_long()
        t_a2.record()
    with torch.cuda.stream(Bs):
        if kind.startswith('torch'):
            Bs.wait_event(e)
        else:
            ops.sync_wait(200)
        busy_short()
        t_b1.record()
    torch.cuda.synchronize()
    return t_a1.elapsed_time(t_b1) * 1e3, t_a1.elapsed_time(t_a2) * 1e3

for kind in ('torch', 'torch_timing', 'pool'):
    for third in (False, True):
        r = [trial(kind, third) for _ in range(5)][1:]
        print(f'{kind:13s} third stream busy={third}:  b1 done after a1 by', ' '.join(f'{a:7.1f}' for a, _ in r), 'us   (a2 lasts', f'{r[0][1]:.0f} us)')
