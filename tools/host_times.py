"""Host (enqueue) time of the phases of one training iteration against their GPU time, bench shape.

The host time is what the launch thread spends in Python + ctypes + HIP runtime to queue the phase; a phase whose
host time approaches its GPU time is launch-bound on a loaded host."""
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
acc = {}
P = time.perf_counter
for it in range(10):
    torch.cuda.synchronize()
    t0 = P(); input_, target = trainer.preprocess(*data); t1 = P()
    trainer.optimizer.zero_grad(); pred = trainer.model(input_); t2 = P()
    m = {}; loss = trainer.compute_loss(pred, target, metrics=m); t3 = P()
    host = trainer._read_back(loss, m, None); t4 = P()
    loss.backward(); t5 = P()
    trainer.optimizer.step(); t6 = P()
    torch.cuda.synchronize(); t7 = P()
    for k, v in (('preprocess', t1 - t0), ('forward', t2 - t1), ('loss', t3 - t2), ('readback(wait fwd)', t4 - t3),
                 ('backward', t5 - t4), ('sgd', t6 - t5), ('final wait', t7 - t6), ('total', t7 - t0)):
        acc.setdefault(k, []).append(v * 1e3)
print('host ms:', {k: round(float(np.median(v)), 3) for k, v in acc.items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(5): trainer.train_one_iteration('train', *data)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
