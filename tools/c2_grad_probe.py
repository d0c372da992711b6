import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import _gradcheck
from oracle import wesup_oracle as orc
from wesup_amd import synth, ops
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
torch.set_num_threads(16)
d = torch.device('cuda:0')
B, H, W, g = 4, 480, 480, 24
weights = orc.make_weights(0, feat_scale=1.0)
imgs, labs, pts, pix = synth.make_batch(5, B, H, W, g)
def run(tag, sk, relu_store, off_chain=True):
    ops.set_streamk(everything=sk)
    tr = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    tr.optimizer, _ = tr.get_default_optimizer(); tr.metric_funcs = [accuracy, dice]; tr.model.train(); tr.tracker.train()
    tr.model._ensure_engine(); tr.model.engine.relu_on_store = relu_store; tr.model.engine.head_wgrad_off_chain = off_chain
    tr.train_one_iteration('train', torch.from_numpy(imgs).to(d), torch.from_numpy(pix).to(d), torch.from_numpy(pts).to(d), torch.from_numpy(labs).to(d))
    ys = _gradcheck.gpu_preactivations(tr.model.engine)
    return tr, ys
tr, ys = run('plain', False, True)
hs = _gradcheck.gpu_mlp_outputs(tr.model.engine)
_, g64, named = _gradcheck.forced_step(weights, imgs, labs, pts, ys, torch.float64, mlp_gpu=hs)
print([n for n in named if n['kind'] == 'fc-relu'][:6])
print('named', len(named), 'bad', sum(not n['near_tie'] for n in named))
def errs(tr, tag):
    out = []
    for k, ref in g64.items():
        sc = float(ref.abs().max()); got = tr.model._grad_views[k].double().cpu()
        out.append((float((got - ref).abs().max()) / sc, k))
    out.sort(reverse=True)
    print(tag, [(f'{e:.2e}', k) for e, k in out[:5]])
errs(tr, 'plain+relu_store')
