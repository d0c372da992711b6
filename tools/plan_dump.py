"""The recorded step plan of the bench shape, node by node: stream, kernel / edge -- the schedule the engine's walk produces.

  python tools/plan_dump.py [batch] > gpurun_out/plan_dump.txt
"""
import os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wesup_oracle as orc
from wesup_amd import synth, _lib
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
dev = torch.device('cuda:0')
B, H, W, g = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 480, 480, 24
trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
trainer.optimizer, _ = trainer.get_default_optimizer()
trainer.metric_funcs = [accuracy, dice]
trainer.tracker.train()
imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
data = tuple(torch.from_numpy(a).to(dev) for a in (imgs, pix, pts, labs))
for _ in range(5):
    trainer.train_one_iteration('train', *data)
torch.cuda.synchronize()
st = next(iter(trainer.step_runner().states.values()))
plan, lib = st.plan, _lib.load()
import ctypes
timing = bool(os.environ.get('WESUP_PLAN_TIMING'))
if timing:
    for _ in range(3):
        trainer.train_one_iteration('train', *data)      # replays (the host is not synchronised in between: steady state)
ids = {}
out2 = (ctypes.c_longlong * 2)()
for i in range(plan.size()):
    s = lib.wesup_plan_node_stream(plan.h, i)
    q = ids.setdefault(s, len(ids))
    name = lib.wesup_plan_node_name(plan.h, i).decode()
    name = re.sub(r'^_Z\d+', '', name)
    t = ''
    if timing and lib.wesup_plan_node_host_ns(plan.h, i, out2) == 0:
        t = f'  issued at {out2[1] / 1e3:8.1f} us, took {out2[0] / 1e3:7.1f} us'
    print(f'{i:4d} s{q} {name[:70]:70s}{t}')
print('cuts at', [c for c, _ in plan.cuts], 'kernels', lib.wesup_plan_kernels(plan.h))
