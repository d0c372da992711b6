"""Peak device memory of the training step at a bench shape:  python tools/mem_probe.py --size 480 --grid 24 --batch 4"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=480); ap.add_argument('--grid', type=int, default=24); ap.add_argument('--batch', type=int, default=4)
a = ap.parse_args()
dev = torch.device('cuda:0')
tr = initialize_trainer('wesup', device='cuda:0', max_superpixels=a.grid ** 2)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
tr.optimizer, _ = tr.get_default_optimizer()
tr.metric_funcs = [accuracy, dice]
tr.tracker.train()
imgs, labs, pts, pix = synth.make_batch(1, a.batch, a.size, a.size, a.grid)
data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
for _ in range(3):
    tr.train_one_iteration('train', *data)
torch.cuda.synchronize()
eng = tr.model.engine
b = eng._last
v = sum(t.numel() * 4 for t in b.V if t is not None)
print(f'B={a.batch} {a.size}x{a.size}, {a.grid ** 2} SP: peak allocated {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB, reserved '
      f'{torch.cuda.memory_reserved() / 2 ** 30:.2f} GiB; kept transformed inputs V {v / 2 ** 30:.2f} GiB')
