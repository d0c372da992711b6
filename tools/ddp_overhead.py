"""Where does the data-parallel path spend its extra time on ONE rank?  Same step: no reducer / reducer without the
collective / with the RCCL all-reduce / with the all-reduce issued from a side thread-free variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer
from wesup_amd.utils.metrics import accuracy, dice
dev = torch.device('cuda:0')
B, H, W, g = 4, 480, 480, 24
imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
def make(mode):
    tr = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g, force_allreduce=(mode == 'rccl'))
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
    tr.optimizer, _ = tr.get_default_optimizer()
    tr.metric_funcs = [accuracy, dice]
    tr.tracker.train()
    if mode != 'none':
        tr.enable_data_parallel()
    return tr
from wesup_amd import ddp as _ddp
_host = {'t': 0.0, 'n': 0, 'fin': 0.0}
_orig_flush, _orig_finish = _ddp.GradAllReducer._flush, _ddp.GradAllReducer.finish
def _flush(self):
    t0 = time.perf_counter(); _orig_flush(self); _host['t'] += time.perf_counter() - t0; _host['n'] += 1
def _finish(self):
    t0 = time.perf_counter(); r = _orig_finish(self); _host['fin'] += time.perf_counter() - t0; return r
_ddp.GradAllReducer._flush, _ddp.GradAllReducer.finish = _flush, _finish
for mode in ('none', 'INIT', 'rccl', 'none', 'rccl', 'none'):
    if mode == 'INIT':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
        print('-- process group initialised')
        continue
    tr = make(mode)
    for _ in range(6):
        tr.train_one_iteration('train', *data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_one_iteration('train', *data)
    torch.cuda.synchronize()
    print(f'{mode:13s}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step   host in _flush {1e3 * _host["t"] / 26:.3f} ms/step over {_host["n"] / 26:.1f} calls, in finish() {1e3 * _host["fin"] / 26:.3f} ms/step')
    _host.update(t=0.0, n=0, fin=0.0)
    del tr
dist.destroy_process_group()
