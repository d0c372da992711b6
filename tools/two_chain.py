"""Does running the VGG conv chain as two independent half-batch chains on two streams beat one full-batch chain?
(two NT kernels in flight: blocks of different launches are not in lock step)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER

d = torch.device('cuda:0')
H = W = 480


def make(B):
    bufs = []
    h, w = H, W
    x0 = torch.randn(B, h, w, 4, device=d)
    for l, (ci, co) in enumerate(CONV_CH):
        y = torch.empty(B, h, w, co, device=d)
        yp = None
        if POOL_AFTER[l]:
            h, w = h // 2, w // 2
            yp = torch.empty(B, h, w, co, device=d)
        bufs.append((y, yp))
    return x0, bufs


wts = []
for l, (ci, co) in enumerate(CONV_CH):
    wt = torch.randn(co, ci, 3, 3, device=d) * 0.02
    wf, _ = ops.pack_conv3x3_weight(wt, need_dgrad=False)
    wts.append((wf, torch.randn(co, device=d)))


def chain(x0, bufs):
    cur = x0
    for l, (ci, co) in enumerate(CONV_CH):
        y, yp = bufs[l]
        ops.conv3x3_fwd(cur, wts[l][0], wts[l][1], co, relu_in=(l > 0), out=y)
        if yp is not None:
            ops.maxpool2_fwd(y, yp)
            cur = yp
        else:
            cur = y


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = make(4)
print('one chain, B=4: %.3f ms' % timeit(lambda: chain(*full)))
ha, hb = make(2), make(2)
s2 = torch.cuda.Stream()


def two():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    chain(*ha)
    with torch.cuda.stream(s2):
        chain(*hb)
    main.wait_stream(s2)


print('two chains, B=2 each, two streams: %.3f ms' % timeit(two))
print('one chain, B=2: %.3f ms (x2 = serial halves)' % timeit(lambda: chain(*ha)))
