"""Trains the bench model for N steps twice on the same synthetic batches -- once with every conv on the direct
implicit-GEMM kernels, once with the Winograd routing -- and prints the two loss curves side by side: the routing changes
the summation order of the convolutions, not what is learned."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, H, W, g = 4, 480, 480, 24
dev = torch.device('cuda:0')
pool = []
for i in range(8):
    imgs, labs, pts, pix = synth.make_batch(100 + i, B, H, W, g)
    pool.append((torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev)))
curves = {}
for name, wino in (('direct', False), ('winograd', True)):
    trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
    trainer.optimizer, trainer.scheduler = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.model.train(); trainer.tracker.train()
    trainer.model._ensure_engine()
    trainer.model.engine.conv_winograd = wino
    trainer.model.engine.wgrad_winograd = wino
    for i in range(N):
        trainer.train_one_iteration('train', *pool[i % len(pool)])
    curves[name] = np.array(trainer.tracker.history['loss'])
d, w = curves['direct'], curves['winograd']
for i in range(0, N, max(1, N // 10)):
    print(f'step {i:3d}: direct {d[i]:.6f}  winograd {w[i]:.6f}  diff {w[i] - d[i]:+.2e}')
print(f'first loss {d[0]:.6f}, last loss direct {d[-1]:.6f} winograd {w[-1]:.6f}; max |diff| over {N} steps {np.abs(w - d).max():.2e} '
      f'(relative {np.abs(w - d).max() / np.abs(d).max():.2e})')
