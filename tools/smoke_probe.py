"""smoke()'s 64x64 case under the three conv routings (direct / Winograd wgrad / Winograd everything): loss, the five
worst gradients against the oracle's raw fp32 gradients, and the fp64 check under the GPU's own discrete decisions
(tests/_gradcheck.py).  Shows that a 1e-2 gradient difference against the raw oracle is ONE near-tie ReLU decision."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
import _gradcheck
B, H, W, g = 2, 64, 64, 6
weights = orc.make_weights(3, feat_scale=0.03)
imgs, labs, pts, pix = synth.make_batch(9, B, H, W, g)
ref_loss, ref_grads, ref_new, _, _, _ = orc.train_step(weights, imgs, labs.astype(np.int64), pts.astype(np.int64))
for cw, ww in ((False, False), (False, True), (True, True)):
    trainer = initialize_trainer('wesup', device='cuda:0')
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    trainer.optimizer, trainer.scheduler = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.tracker.train()
    trainer.model._ensure_engine()
    trainer.model.engine.conv_winograd = cw
    trainer.model.engine.wgrad_winograd = ww
    trainer.train_one_iteration('train', torch.from_numpy(imgs), torch.from_numpy(pix).long(), torch.from_numpy(pts).long(), torch.from_numpy(labs))
    loss = trainer.tracker.history['loss'][0]
    errs = {}
    for k, gr in ref_grads.items():
        a = trainer.model._grad_views[k].double().cpu()
        errs[k] = float((a - gr.double()).abs().max() / (gr.double().abs().max() + 1e-30))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print('conv_winograd', cw, 'wgrad_winograd', ww, 'loss', loss, ref_loss, 'worst', top, flush=True)
    try:
        worst, n = _gradcheck.check_gradients(trainer.model, weights, imgs, labs.astype(np.int64), pts.astype(np.int64))
        print('   gradcheck under GPU decisions: worst', worst, 'named near-ties', n, flush=True)
    except AssertionError as e:
        print('   gradcheck FAILED:', str(e)[:600], flush=True)
