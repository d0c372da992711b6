"""cProfile of the launch thread over training iterations at batch 1 (host-bound there): python tools/host_profile.py [B]"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
dev = torch.device('cuda:0')
B, H, W, g = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 480, 480, 14
trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
trainer.optimizer, _ = trainer.get_default_optimizer()
trainer.metric_funcs = [accuracy, dice]
trainer.tracker.train()
imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
for _ in range(5): trainer.train_one_iteration('train', *data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): trainer.train_one_iteration('train', *data)
torch.cuda.synchronize()
print('%.3f ms per iteration' % ((time.perf_counter() - t0) / 30 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): trainer.train_one_iteration('train', *data)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
