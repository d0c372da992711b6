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


def _make(weights, device='cuda:0'):
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    tr = initialize_trainer('wesup', device=device)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    tr.optimizer, _ = tr.get_default_optimizer()
    tr.metric_funcs = [accuracy, dice]
    tr.tracker.train()
    return tr


def _batch():
    from wesup_amd import synth
    return synth.make_batch(21, 4, 64, 64, 6)


def _worker(rank, world, port, out, backend='gloo'):
    import torch.distributed as dist
    from oracle import wesup_oracle as orc
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = 'cuda:0'
    if backend == 'nccl':                                # RCCL: one rank per device
        dev = f'cuda:{rank}'
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        tr = _make(orc.make_weights(5, feat_scale=0.03), dev)
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
    _two_ranks_equal_one_rank('gloo')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank: runs on a box with >= 2 GPUs')
def test_rccl_two_ranks_equal_one_rank():
    """The same statement with RCCL carrying the exchange (backend "nccl", one rank per device, fresh processes): the
    asynchronous bucketed all-reduces launched from the wgrad and side streams during backward, finish() ordering the
    optimiser behind them.  Skipped on the one-GPU test box; a multi-GPU box picks it up under `pytest -m gpu`."""
    _two_ranks_equal_one_rank('nccl')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs one device per rank')
def test_bench_two_ranks_over_rccl_reports_the_collective():
    """`python bench.py --gpus 2` on two devices: RCCL ranks, and the line carries the measured exposed all-reduce time and
    the bucket launch offsets (overlap as a measurement, not a claim)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
                        '--no-cpu-baseline', '--no-kernel-timing'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    col = out['collective']
    assert out['n_gpus'] == 2 and col['ranks'] == 2 and col['backend'] == 'nccl'
    assert col['exposed_ms'] is not None and col['exposed_ms']['median'] >= 0 and len(col['buckets']['bytes']) >= 2
    assert sum(col['buckets']['bytes']) >= 18868194 * 4


def _two_ranks_equal_one_rank(backend):
    import torch.multiprocessing as mp
    from oracle import wesup_oracle as orc
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, backend)) for r in range(2)]
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


def _nan_worker(rank, world, port, out):
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
        img = torch.from_numpy(imgs[sl]).clone()
        if rank == 1:
            img[0, 0, 3, 3] = float('nan')              # ONE rank sees a NaN loss
        before = tr.model._flat.clone()
        raised = False
        try:
            tr.train_one_iteration('train', img, torch.from_numpy(pix[sl]).long(), torch.from_numpy(pts[sl]).long(),
                                   torch.from_numpy(labs[sl]))
        except ValueError as ex:
            raised = 'nan' in str(ex).lower()
        torch.cuda.synchronize()
        untouched = bool(torch.equal(before, tr.model._flat))
        # the next iteration on clean data runs normally on both ranks (no stale reducer state)
        tr.train_one_iteration('train', torch.from_numpy(imgs[sl]), torch.from_numpy(pix[sl]).long(),
                               torch.from_numpy(pts[sl]).long(), torch.from_numpy(labs[sl]))
        torch.cuda.synchronize()
        moved = not bool(torch.equal(before, tr.model._flat))
        out.put((rank, raised, untouched, moved, float(tr.model._flat.double().sum())))
    finally:
        dist.destroy_process_group()


def test_nan_on_one_rank_stops_every_rank_before_the_update():
    """models/base.py:202-203 under data parallelism: the rank with the NaN loss AND its peer raise ValueError before
    optimizer.step (the peer's own loss is finite, but the all-reduced gradients are not), nobody's weights move, and
    the next clean iteration leaves both ranks with identical parameters."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nan_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=280) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, raised, untouched, moved, _ in res:
        assert raised and untouched and moved, res
    assert res[0][4] == res[1][4]                       # replicas still agree bit for bit


def test_bench_self_launch_two_ranks_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns both ranks (here on the one card, gloo
    carrying the exchange), rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rehearse-on-one-gpu', '--steps', '3',
                        '--warmup', '1', '--batch', '1', '--size', '64', '--grid', '4', '--no-cpu-baseline',
                        '--no-kernel-timing'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 2 and out['config']['parallelism'] == 'dp2'
    assert out['collective']['ranks'] == 2 and out['rank_time']['max_s'] >= out['rank_time']['min_s'] > 0
    assert out['value'] > 0 and 'custom shape' in out['config']['workload']


def _fit_worker(rank, world, port, root, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      RECORD_ROOT=root)
    from wesup_amd.train import fit
    trainer = fit('synthetic:64:64:4:8', model='wesup', epochs=2, batch_size=2, num_workers=0, dist_backend='gloo', no_val=True)
    import torch.distributed as dist
    out.put((rank, trainer.world_size, str(trainer.record_dir), float(trainer.model._flat.double().sum()),
             len(trainer.tracker.history['loss'])))
    dist.destroy_process_group()


def test_fit_under_a_multi_process_launcher(tmp_path):
    """`python -m torch.distributed.run ... -m wesup_amd.train DATA`: fit() sees WORLD_SIZE, joins the process group and
    trains data-parallel -- one record directory (rank 0's), half of the 8 items per rank and epoch, replicas
    bit-identical after two epochs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=280) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, w0, d0, s0, n0), (r1, w1, d1, s1, n1) = res
    assert (w0, w1) == (2, 2) and d0 == d1 and s0 == s1
    assert n0 == n1 == 2                                   # 8 items / 2 ranks / batch 2 = 2 iterations in the last epoch
    rd = [p for p in tmp_path.iterdir()]
    assert len(rd) == 1 and (rd[0] / 'history.csv').exists() and len(list((rd[0] / 'checkpoints').glob('*.pth'))) == 1
