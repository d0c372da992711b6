import sys, os
sys.path.insert(0, '/root/repo')
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
for mode in (True, False, True, False):
    trainer.model._ensure_engine()
    trainer.model.engine.two_streams = mode
    for _ in range(4): trainer.train_one_iteration('train', *data)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        input_, target = trainer.preprocess(*data)
        trainer.optimizer.zero_grad()
        e0.record()
        pred = trainer.model(input_)
        e1.record()
        loss = trainer.compute_loss(pred, target, metrics={})
        loss.backward()
        e2.record()
        trainer.optimizer.step()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    print('streams' if mode else 'single stream', 'forward %.3f ms  loss+backward %.3f ms' % tuple(np.median(np.array(ts), 0)))
