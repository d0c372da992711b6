"""The product + output-transform kernel of conv1_2's input gradient (480^2, 64 -> 64, B = 4) alone, by epilogue variant:
the transformed operand is prepared once (v_pre), only the one-kernel product route is timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, synth
d = torch.device('cuda:0')
B, H, W, C, g = 4, 480, 480, 64, 24
labs = np.stack([synth.voronoi_labels(b, H, W, g) for b in range(B)])
K = int(labs.max()) + 1
meta = ops.sp_preprocess(torch.from_numpy(labs).to(d), None, K)
dy = torch.randn(B, H, W, C, device=d)
y0 = torch.randn(B, H, W, C, device=d)
w = torch.randn(C, C, 3, 3, device=d) * 0.05
_, ud = ops.winograd_pack_weight(w, need_fwd=False, m=4)
V = ops.winograd_input_transform(dy, m=4)
bits = torch.zeros(B, H, W, C // 4, dtype=torch.uint8, device=d)
ops.conv3x3_fwd_winograd(y0, ops.winograd_pack_weight(w, need_dgrad=False, m=4)[0], None, relu_in=True, m=4, relu_bits_out=bits)
side = torch.randn(B, K, C, device=d)
out = torch.zeros(B, H, W, C, device=d)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (
    ('plain store', lambda: ops.conv3x3_dgrad_winograd(dy, ud, out=out, m=4, v_pre=V)),
    ('float mask', lambda: ops.conv3x3_dgrad_winograd(dy, ud, mask_src=y0, out=out, m=4, v_pre=V)),
    ('bit mask', lambda: ops.conv3x3_dgrad_winograd(dy, ud, out=out, m=4, v_pre=V, mask_bits=bits)),
    ('float mask + accumulate', lambda: ops.conv3x3_dgrad_winograd(dy, ud, mask_src=y0, out=out, accumulate=True, m=4, v_pre=V)),
    ('bit mask + accumulate', lambda: ops.conv3x3_dgrad_winograd(dy, ud, out=out, accumulate=True, m=4, v_pre=V, mask_bits=bits)),
    ('float mask + gather', lambda: ops.conv3x3_dgrad_winograd_gather(dy, ud, side, meta.new_row, meta.area_new, out=out, mask_src=y0, v_pre=V)),
    ('bit mask + gather', lambda: ops.conv3x3_dgrad_winograd_gather(dy, ud, side, meta.new_row, meta.area_new, out=out, mask_bits=bits, v_pre=V)),
    ('plain store (again)', lambda: ops.conv3x3_dgrad_winograd(dy, ud, out=out, m=4, v_pre=V)),
    ('bit mask (again)', lambda: ops.conv3x3_dgrad_winograd(dy, ud, out=out, m=4, v_pre=V, mask_bits=bits)),
):
    print(f'{name:28s} {timeit(fn):7.1f} us')
