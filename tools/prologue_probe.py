"""Steady-state gap between the end of one step (event behind the SGD kernel) and the start of conv1_1 of the next, and the
GPU time from there to the end of forward / backward: is the head of the step launch-bound without a profiler attached?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth, ops
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
dev = torch.device('cuda:0')
B, H, W, g = 4, 480, 480, 24
trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
trainer.optimizer, _ = trainer.get_default_optimizer()
trainer.metric_funcs = [accuracy, dice]
trainer.tracker.train()
imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
for _ in range(5): trainer.train_one_iteration('train', *data)
torch.cuda.synchronize()
ev = {}
def E(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(tag, []).append(e)
orig_conv, orig_bwd, orig_loss = ops.conv3x3_fwd, trainer.model.engine.backward, trainer.compute_loss
def conv(*a, **k):
    E('conv1_1')
    return orig_conv(*a, **k)
def loss(*a, **k):
    E('fwd_end')
    return orig_loss(*a, **k)
ops.conv3x3_fwd = conv
trainer.compute_loss = loss
N = 12
for i in range(N):
    E('begin')
    trainer.train_one_iteration('train', *data)
E('begin')
torch.cuda.synchronize()
b, c, f = ev['begin'], ev['conv1_1'], ev['fwd_end']
gap = [b[i].elapsed_time(c[i]) for i in range(2, N)]
fwd = [c[i].elapsed_time(f[i]) for i in range(2, N)]
rest = [f[i].elapsed_time(b[i + 1]) for i in range(2, N)]
tot = [b[i].elapsed_time(b[i + 1]) for i in range(2, N)]
print('ms  step-begin -> conv1_1 launch point: median %.3f (min %.3f max %.3f)' % (np.median(gap), min(gap), max(gap)))
print('    conv1_1 -> forward end %.3f;  forward end -> step end (loss, backward, SGD) %.3f;  step %.3f' % (np.median(fwd), np.median(rest), np.median(tot)))
