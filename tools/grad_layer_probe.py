"""Diagnostic: per-layer activations / gradients of one HIP training step vs an fp64 evaluation of the oracle;
prints where they differ (used to show that the only outliers are ReLU sign flips at pre-activations within fp32
rounding of zero)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
H, W = 70, 50
imgs = np.stack([synth.synth_image(40 + b, H, W) for b in range(2)])
segs = np.stack([synth.voronoi_labels(50, H, W, 5), synth.voronoi_labels(51, H, W, 3)])
masks = np.stack([synth.point_mask(60 + b, segs[b], 0.4, 2, tie_every=2) for b in range(2)])
pix = np.stack([synth.pixel_mask(70 + b, H, W) for b in range(2)])
weights = orc.make_weights(7, feat_scale=0.03)
w = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in weights.items()}
x = torch.from_numpy(imgs).double()
ys = []
orig = orc.backbone_side_outputs
def patched(wd, xx):
    outs, h = [], xx
    for li, (idx, off) in enumerate(zip(orc.CONV_IDX, orc.SIDE_OFF)):
        y = F.conv2d(h, wd[f'backbone.{idx}.weight'], wd[f'backbone.{idx}.bias'], padding=1)
        y.retain_grad(); ys.append(y)
        outs.append(F.conv2d(y, wd[f'side_conv{off}.weight'], wd[f'side_conv{off}.bias']))
        h = F.relu(y)
        if orc.POOL_AFTER[li] and li != 12:
            h = F.max_pool2d(h, 2, 2)
    return outs
orc.backbone_side_outputs = patched
losses = []
for b in range(2):
    o = orc.forward_image(w, x[b], torch.from_numpy(segs[b].astype(np.int64)), torch.from_numpy(masks[b].astype(np.int64)))
    losses.append(orc.compute_loss(o['sp_pred'], o['sp_features'], o['pp']['sp_labels']))
(torch.stack(losses).mean()).backward()
d = torch.device('cuda:0')
tr = initialize_trainer('wesup', device='cuda:0')
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
tr.optimizer, _ = tr.get_default_optimizer(); tr.metric_funcs = [accuracy, dice]; tr.tracker.train()
tr.train_one_iteration('train', torch.from_numpy(imgs).to(d), torch.from_numpy(pix).long().to(d), torch.from_numpy(masks).long().to(d), torch.from_numpy(segs))
torch.cuda.synchronize()
bufs = list(tr.model.engine._bufs.values())[-1]
for l in range(13):
    yref = torch.stack([ys[l], ys[13 + l]])[:, 0]              # (B,C,h,w) one entry per image
    gref = torch.stack([ys[l].grad, ys[13 + l].grad])[:, 0]
    ye = bufs.y[l].double().cpu().permute(0, 3, 1, 2)
    ge = bufs.G[l].double().cpu().permute(0, 3, 1, 2)
    ey = float((ye - yref).abs().max() / yref.abs().max())
    diff = (ge - gref).abs()
    eg = float(diff.max() / gref.abs().max())
    idx = np.unravel_index(int(diff.argmax()), diff.shape)
    print(f'layer {l:2d} {tuple(yref.shape)}  y err {ey:.2e}  G err {eg:.2e}  worst at (b,c,h,w)={idx}  rows with err>1e-5*max: '
          f'{sorted(set(np.nonzero((diff.amax(dim=(0,1,3)) > 1e-5 * float(gref.abs().max())).numpy())[0].tolist()))[:12]} cols: '
          f'{sorted(set(np.nonzero((diff.amax(dim=(0,1,2)) > 1e-5 * float(gref.abs().max())).numpy())[0].tolist()))[:12]}')
l = 4
yref = torch.stack([ys[l], ys[13 + l]])[:, 0]; gref = torch.stack([ys[l].grad, ys[13 + l].grad])[:, 0]
ye = bufs.y[l].double().cpu().permute(0, 3, 1, 2); ge = bufs.G[l].double().cpu().permute(0, 3, 1, 2)
diff = (ge - gref).abs()
bad = torch.nonzero(diff > 1e-4 * gref.abs().max())
print('bad elements in G[4]:', bad.shape[0])
for (b_, c_, h_, w_) in bad[:8].tolist():
    print(f'  (b={b_},c={c_},h={h_},w={w_}): y_ref {yref[b_,c_,h_,w_].item():+.3e} y_gpu {ye[b_,c_,h_,w_].item():+.3e}   G_ref {gref[b_,c_,h_,w_].item():+.3e} G_gpu {ge[b_,c_,h_,w_].item():+.3e}')
