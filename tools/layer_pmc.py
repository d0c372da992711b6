"""Driver for per-layer PMC collection: every conv3x3 layer of the bench shape (B=4, 480x480) x {fwd, dgrad, wgrad}
(as the engine routes them -- wesup_amd.engine.default_route: implicit GEMM below 128 input channels, Winograd F(4x4,3x3)
domain from there up),
each launched `reps` times in a fixed order, with a manifest of that order so that tools/roofline_inputs.py can map
the dispatches of the rocprofv3 counter file back to (layer, pass).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d OUT -- python3 tools/layer_pmc.py MANIFEST.json
"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wesup_amd import ops
from wesup_amd.engine import CONV_CH, POOL_AFTER, default_route

d = torch.device('cuda:0')
B, H, W = 4, 480, 480
reps = 3
manifest = {'shape': [B, H, W], 'reps': reps, 'order': []}     # order: [layer, pass, flop] per GEMM launch group
h, w = H, W
for l, (ci, co) in enumerate(CONV_CH):
    cin = 4 if l == 0 else ci
    x = torch.randn(B, h, w, cin, device=d)
    wt = torch.randn(co, ci, 3, 3, device=d) * 0.02
    bias = torch.randn(co, device=d)
    dy = torch.randn(B, h, w, co, device=d)
    wf, wd = ops.pack_conv3x3_weight(wt, need_dgrad=(l > 0))
    y = torch.empty(B, h, w, co, device=d)
    dx = torch.empty(B, h, w, cin, device=d)
    dw = torch.empty(co, ci, 3, 3, device=d)
    db = torch.empty(co, device=d)
    m = default_route(ci, co, h, w, B) if ci >= 32 else 0      # the engine's routing
    wino = m > 0
    # FLOPs the MFMA pipe executes in the main GEMM launch of the group (Winograd: (m+2)^2 positions x m-by-m tiles)
    fl = 2.0 * ops.winograd_positions(m) * ops.winograd_tiles(B, h, w, m) * ci * co if wino else 2.0 * B * h * w * ci * co * 9
    if wino:
        uf, ud = ops.winograd_pack_weight(wt, m=m)
        v_keep = torch.empty(ops.winograd_positions(m), ops.winograd_tiles(B, h, w, m), ci, device=d)
    for _ in range(reps):
        if wino:
            ops.conv3x3_fwd_winograd(x, uf, bias, False, out=y, v_keep=v_keep, m=m)
        else:
            ops.conv3x3_fwd(x, wf, bias, co, relu_in=False, out=y)
    manifest['order'].append([l, 'fwd', fl])
    if l > 0:
        for _ in range(reps):
            if wino:
                ops.conv3x3_dgrad_winograd(dy, ud, mask_src=x, out=dx, accumulate=True, m=m)
            else:
                ops.conv3x3_dgrad(dy, wd, ci, mask_src=x, out=dx, accumulate=True)
        manifest['order'].append([l, 'dgrad', fl])
    for _ in range(reps):
        if wino:
            ops.conv3x3_wgrad_winograd(x, dy, False, dw=dw, db=db, v_pre=v_keep, m=m)
        else:
            ops.conv3x3_wgrad(x, dy, ci, relu_in=False, dw=dw, db=db)
    manifest['order'].append([l, 'wgrad', fl])
    torch.cuda.synchronize()
    if POOL_AFTER[l]:
        h, w = h // 2, w // 2
    del x, wt, dy, y, dx
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(manifest, f)
print('layer_pmc done:', len(manifest['order']), 'launch groups')
