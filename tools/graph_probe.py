"""Feasibility probe: one training iteration (preprocess + forward + loss + backward, without the host read-back and the
optimiser) captured into a HIP graph through torch.cuda.CUDAGraph and replayed, against eager launches, at batch 1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import synth, ops
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
data = [torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev)]
for _ in range(5): trainer.train_one_iteration('train', *data)
torch.cuda.synchronize()


def body():
    input_, target = trainer.preprocess(*data)
    trainer.optimizer.zero_grad()
    pred = trainer.model(input_)
    loss = trainer.compute_loss(pred, target, metrics={})
    loss.backward()
    return loss


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('eager, no optimiser: %.3f ms per iteration' % timeit(lambda: body()))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr, stream=s):
        loss = body()
    torch.cuda.synchronize()
    print('captured')
    print('graph replay: %.3f ms per iteration' % timeit(lambda: gr.replay()))
    print('loss after replay', float(loss))
except Exception as ex:
    import traceback; traceback.print_exc()
    print('capture failed:', type(ex).__name__, str(ex)[:300])
