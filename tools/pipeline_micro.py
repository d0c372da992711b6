"""End-to-end step from decoded uint8 batches: GPU augmentation + GPU SLIC + training step (bench shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import wesup_oracle as orc
from wesup_amd import ops, synth
from wesup_amd.utils import data as D
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
dev = torch.device('cuda:0')
B, H, W = 4, 480, 480
rs = np.random.RandomState(0)
imgs = np.ascontiguousarray((np.stack([synth.synth_image(i, H, W) for i in range(B)]).transpose(0, 2, 3, 1) * 255).astype(np.uint8))
mask = (rs.random_sample((B, H, W)) > 0.5).astype(np.uint8)
d_img, d_mask = torch.from_numpy(imgs).to(dev), torch.from_numpy(mask).to(dev)
params = torch.from_numpy(np.stack([D.sample_params(rs, H, W, True)[0] for _ in range(B)])).to(dev)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print('wesup_augment, batch of 4 x 480x480: %.1f us' % (t(lambda: ops.augment(d_img, d_mask, params)) * 1e3))
trainer = initialize_trainer('wesup', device='cuda:0')           # sp_area 200 -> ~1150 SLIC segments per image
trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
trainer.optimizer, _ = trainer.get_default_optimizer()
trainer.metric_funcs = [accuracy, dice]
trainer.tracker.train()
pts = torch.zeros(B, 2, H, W, dtype=torch.uint8, device=dev)
idx = rs.randint(0, H, (B, 120, 2))
for b in range(B):
    pts[b, rs.randint(0, 2, 120), idx[b, :, 0], idx[b, :, 1]] = 1
def step():
    img, pm = ops.augment(d_img, d_mask, params)
    trainer.train_one_iteration('train', img, pm, pts)          # no label map given: GPU SLIC inside preprocess
ms = t(step, 10)
print('augment + GPU SLIC (sp_area 200) + training step: %.2f ms/step = %.1f img/s' % (ms, B / ms * 1e3))

# SLIC-ahead (utils/data.py DevicePrefetcher with segment_fn): augmentation + SLIC of the NEXT batch on a second stream,
# beside the training step of the current one; the superpixel counts reach the host through pinned memory in the
# meantime, so the step pads to the exact row count instead of the worst-case bound
side = torch.cuda.Stream(device=dev)
seg_fn = trainer.prefetch_segment_fn()
state = {}
def stage():
    with torch.cuda.stream(side):
        img, pm = ops.augment(d_img, d_mask, params)
        seg, n_dev = seg_fn(img)
        counts = torch.empty(n_dev.shape, dtype=n_dev.dtype).pin_memory()
        counts.copy_(n_dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    return img, pm, seg, counts, ev
state['nxt'] = stage()
def step_ahead():
    img, pm, seg, counts, ev = state['nxt']
    state['nxt'] = stage()
    torch.cuda.current_stream().wait_event(ev)
    ev.synchronize()
    for t_ in (img, pm, seg):
        t_.record_stream(torch.cuda.current_stream())
    trainer.train_one_iteration('train', img, pm, pts, D.LabelMaps(seg, [int(v) for v in counts]))
ms = t(step_ahead, 10)
print('the same with augmentation + SLIC one batch ahead on a second stream, exact counts: %.2f ms/step = %.1f img/s' % (ms, B / ms * 1e3))
print('superpixels per image:', [int(v) for v in state['nxt'][3]])
