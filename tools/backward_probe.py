"""Where the time of one steady-state step goes on the main stream (no profiler attached): events at step begin, conv1_1,
forward end (compute_loss), backward begin (classifier_bwd), first conv dgrad (G of conv5_3 ready), last dgrad, step end."""
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
def wrap(obj, name, tag, first_only=True):
    orig = getattr(obj, name)
    def f(*a, **k):
        if not first_only or len(ev.get(tag, [])) < len(ev['begin']):
            E(tag)
        return orig(*a, **k)
    setattr(obj, name, f)
wrap(ops, 'conv3x3_fwd', 'conv1_1')
wrap(trainer, 'compute_loss', 'fwd_end')
wrap(ops, 'classifier_bwd', 'bwd_begin')
wrap(ops, 'winograd_dual_transform', 'first_dgrad')
wrap(ops, 'upsample_bwd_fused_group', 'x')          # (not on the main stream; ignored)
wrap(trainer.optimizer, 'step', 'bwd_end')
N = 12
for i in range(N):
    E('begin')
    trainer.train_one_iteration('train', *data)
E('begin')
torch.cuda.synchronize()
tags = ['begin', 'conv1_1', 'fwd_end', 'bwd_begin', 'first_dgrad', 'bwd_end']
for a, b_ in zip(tags, tags[1:]):
    d = [ev[a][i].elapsed_time(ev[b_][i]) for i in range(2, N)]
    print(f'{a:12s} -> {b_:12s} median {np.median(d):7.3f} ms  (min {min(d):.3f} max {max(d):.3f})')
d = [ev['bwd_end'][i].elapsed_time(ev['begin'][i + 1]) for i in range(2, N)]
print(f'{"bwd_end":12s} -> next begin   median {np.median(d):7.3f} ms')
d = [ev['begin'][i].elapsed_time(ev['begin'][i + 1]) for i in range(2, N)]
print(f'step median {np.median(d):.3f} ms')
