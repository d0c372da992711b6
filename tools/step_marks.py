"""Where the three streams of ONE untraced training step are at which time: HIP events around chosen launch classes, all
measured against one event at the head of the iteration (rocprofv3's queue interception stretches cross-queue waits --
tools/step_trace.py's caution -- so gaps between streams are read off here, not off a kernel trace).

  python3 tools/step_marks.py [classes,comma,separated | all] [batch]        (default: the backward head's classes, batch 4)
Prints, per timed pair of the LAST of 6 walked iterations: start and end (us from the head of the iteration), stream, class.
Events fence the queue they sit on: with `all` the step itself is ~16 % slower; pick few classes for gap questions.
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.engine import KernelTimer
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice


class MarkTimer(KernelTimer):
    """KernelTimer that keeps the events (and the stream they were recorded on) instead of folding them into class totals."""

    def __init__(self, only):
        super().__init__()
        self.enabled = True
        self.only = only
        self.marks = []

    def begin(self, tag):
        if self.only is not None and tag not in self.only:
            return None
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        return (tag, s, torch.cuda.current_stream().cuda_stream)

    def end(self, tok, work=0.0):
        if tok is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((tok[0], tok[1], e, tok[2]))

    def collect(self):
        return {}


def main():
    sel = sys.argv[1] if len(sys.argv) > 1 else 'mlp_fwd,mlp_bwd,mlp_wgrad,upsample_mat_bwd,side_bwd,winograd_transform,upsample_bwd'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    only = None if sel == 'all' else set(sel.split(','))
    dev = torch.device('cuda:0')
    H, W, g = 480, 480, 24
    trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g, step_plan=False)
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
    trainer.optimizer, _ = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.tracker.train()
    imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
    data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
    for _ in range(3):
        trainer.train_one_iteration('train', *data)
    eng = trainer.model.engine
    eng.timer = T = MarkTimer(only)
    streams = {}
    for it in range(6):
        T.marks = []
        torch.cuda.synchronize()
        head = torch.cuda.Event(enable_timing=True)
        head.record()
        trainer.train_one_iteration('train', *data)
        tail = torch.cuda.Event(enable_timing=True)
        tail.record()
        torch.cuda.synchronize()
    print(f'# batch {B}, classes {sel}; iteration {head.elapsed_time(tail) * 1e3:.0f} us head to tail (with these events in it)')
    rows = []
    for tag, s, e, st in T.marks:
        q = streams.setdefault(st, len(streams) + 1)
        rows.append((head.elapsed_time(s) * 1e3, head.elapsed_time(e) * 1e3, q, tag))
    rows.sort()
    for a, b, q, tag in rows:
        print(f'{a:9.1f} {b:9.1f}  {b - a:7.1f} us  s{q}  {tag}')


if __name__ == '__main__':
    main()
