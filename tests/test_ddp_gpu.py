"""Data parallelism end to end on the GPU kernels: two processes share the one card (gloo carries the exchange, so no
second GPU is needed), each takes half of a batch through the HIP training step with the gradient all-reducer
attached, and the updated parameters must equal those of one process stepping on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(weights):
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    tr = initialize_trainer('wesup', device='cuda:0')
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    tr.optimizer, _ = tr.get_default_optimizer()
    tr.metric_funcs = [accuracy, dice]
    tr.tracker.train()
    return tr


def _batch():
    from wesup_amd import synth
    return synth.make_batch(21, 4, 64, 64, 6)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from oracle import wesup_oracle as orc
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        tr = _make(orc.make_weights(5, feat_scale=0.03))
        tr.enable_data_parallel(bucket_bytes=8 << 20)
        imgs, labs, pts, pix = _batch()
        sl = slice(2 * rank, 2 * rank + 2)
        tr.train_one_iteration('train', torch.from_numpy(imgs[sl]), torch.from_numpy(pix[sl]).long(),
                               torch.from_numpy(pts[sl]).long(), torch.from_numpy(labs[sl]))
        torch.cuda.synchronize()
        if rank == 0:
            out.put(({k: v.detach().cpu().numpy() for k, v in tr.model.state_dict().items()},
                     {k: v.detach().float().cpu().numpy() for k, v in tr.model._grad_views.items()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_whole_batch():
    import torch.multiprocessing as mp
    from oracle import wesup_oracle as orc
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got, got_grads = out.get(timeout=280)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tr = _make(orc.make_weights(5, feat_scale=0.03))
    imgs, labs, pts, pix = _batch()
    tr.train_one_iteration('train', torch.from_numpy(imgs), torch.from_numpy(pix).long(), torch.from_numpy(pts).long(),
                           torch.from_numpy(labs))
    torch.cuda.synchronize()
    want = {k: v.detach().cpu().numpy() for k, v in tr.model.state_dict().items()}
    want_grads = {k: v.detach().float().cpu().numpy() for k, v in tr.model._grad_views.items()}
    for k in want_grads:
        # rank 0's buffer holds the SUM over ranks of the per-rank means (the 1/world lives in the SGD kernel)
        scale = np.abs(want_grads[k]).max()
        err = np.abs(got_grads[k] / 2 - want_grads[k]).max()
        assert err <= 1e-4 * scale + 1e-12, (k, err, scale)
    for k in want:
        # the parameters moved by the same update up to the fp32 resolution of the parameter itself
        assert np.abs(got[k] - want[k]).max() <= 2e-7 * max(np.abs(want[k]).max(), 1e-3), k
