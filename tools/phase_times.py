"""Wall time of the phases of one training iteration (sync after each phase), bench shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
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
for _ in range(3): trainer.train_one_iteration('train', *data)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(10):
    t0 = T(); input_, target = trainer.preprocess(*data); t1 = T()
    trainer.optimizer.zero_grad(); pred = trainer.model(input_); t2 = T()
    m = {}; loss = trainer.compute_loss(pred, target, metrics=m); host = trainer._read_back(loss, m, None); t3 = T()
    loss.backward(); t4 = T()
    trainer.optimizer.step(); t5 = T()
    for k, v in (('preprocess', t1 - t0), ('forward', t2 - t1), ('loss+sync', t3 - t2), ('backward', t4 - t3), ('sgd', t5 - t4)):
        acc.setdefault(k, []).append(v * 1e3)
    # cpu-only enqueue time of a full iteration (no syncs inside)
t0 = T(); trainer.train_one_iteration('train', *data); t_enq = time.perf_counter() - t0; t1 = T()
print({k: round(float(np.median(v)), 3) for k, v in acc.items()}, 'sum', round(sum(float(np.median(v)) for v in acc.values()), 3))
print('one iteration: cpu enqueue+sync point %.2f ms, total %.2f ms' % (t_enq * 1e3, (t1 - t0) * 1e3))
