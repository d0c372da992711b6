"""Data-parallel layer on CPU: world_size-2 gloo processes exercise the bucketed gradient all-reduce
(wesup_amd.ddp.GradAllReducer), the shard sampler and the averaging convention (sum over ranks, 1/world folded
into the SGD step).  No GPU and no HIP kernel is involved: the reducer only sees a flat fp32 buffer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wesup_amd import ddp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout():
    # same shape of problem as the model: head params at the end, backbone at the front, 64-element padding
    names, sizes = [], {}
    for i in range(5):
        names.append(f'backbone.{i}.weight'); sizes[names[-1]] = 640 + 64 * i
    for i in range(3):
        names.append(f'side_conv{i}.weight'); sizes[names[-1]] = 128
    names.append('fc.weight'); sizes['fc.weight'] = 1024
    offs, total = {}, 0
    for n in names:
        offs[n] = total
        total += sizes[n]
    return names, offs, sizes, total


def _worker(rank, world, port, bucket_bytes, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        names, offs, sizes, total = _layout()
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        local = flat.clone()
        red = ddp.GradAllReducer(flat, offs, sizes, bucket_bytes=bucket_bytes)
        # the engine's completion order: head, backbone from the last layer down, side branch last
        red.ready(['fc.weight'])
        for i in range(4, -1, -1):
            red.ready([f'backbone.{i}.weight'])
        red.ready([f'side_conv{i}.weight' for i in (2, 1, 0)])
        launched = red.finish()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = sum(gathered)
        ok = torch.allclose(flat, expect, atol=1e-6)
        covered = sorted(launched)
        full = covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        # broadcast_parameters-equivalent on a plain tensor
        p = torch.full((8,), float(rank))
        dist.broadcast(p, src=0)
        if rank == 0:
            out.put((ok, full, len(launched), float(p.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [1 << 30, 4096, 1])
def test_bucketed_allreduce_world2(bucket_bytes):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, full, n, psum = out.get(timeout=10)
    assert ok, 'all-reduced flat gradient != sum over ranks'
    assert full, 'buckets do not tile the flat buffer exactly once'
    assert psum == 0.0
    if bucket_bytes == 1 << 30:
        assert n <= 3          # contiguous ranges coalesce: [fc], [backbone...], [side...] or fewer
    if bucket_bytes == 1:
        assert n >= 7


def test_reducer_single_process_is_a_noop():
    names, offs, sizes, total = _layout()
    flat = torch.arange(total, dtype=torch.float32)
    ref = flat.clone()
    red = ddp.GradAllReducer(flat, offs, sizes)
    red.ready(names[::-1])
    done = red.finish()
    assert torch.equal(flat, ref)
    assert sum(b - a for a, b in done) == total


def test_shard_indices_partition_the_dataset():
    for n, world in [(10, 2), (17, 4), (8, 8), (5, 8)]:
        seen = []
        for r in range(world):
            idx = ddp.shard_indices(n, r, world, seed=3, epoch=1)
            assert len(idx) == (n + world - 1) // world
            seen += idx
        assert set(seen) == set(range(n))
        assert ddp.shard_indices(n, 0, world, seed=3, epoch=1) == ddp.shard_indices(n, 0, world, seed=3, epoch=1)
        assert ddp.shard_indices(max(n, 9), 0, world, seed=3, epoch=1) != ddp.shard_indices(max(n, 9), 0, world, seed=3, epoch=2)


def test_weight_bias_pairs_merge_into_one_range():
    """The engine reports (weight, bias) of a layer in one call while backward walks the flat buffer downwards: the
    pair must extend the pending range, not flush it (28 flushes per step instead of 6 buckets)."""
    names, offs, sizes, total = [], {}, {}, 0
    for i in range(6):
        for t, n in (('weight', 640), ('bias', 64)):
            k = f'backbone.{i}.{t}'
            names.append(k); offs[k] = total; sizes[k] = n; total += n
    flat = torch.zeros(total)
    red = ddp.GradAllReducer(flat, offs, sizes, bucket_bytes=1 << 30)
    for i in range(5, -1, -1):
        red.ready([f'backbone.{i}.weight', f'backbone.{i}.bias'])
        assert red.launched == []                    # nothing flushed: one growing contiguous range
    done = red.finish()
    assert done == [(0, total)]


# ------------------------------------------------------------------ world size 8 on the model's real parameter layout
def _model_layout():
    """Names, 64-element-aligned offsets and padded sizes of the 18 868 194 parameters, as WESUP._ensure_engine lays them
    out, and the order in which the engine's backward reports them (engine.py: head, side convs deepest first, backbone
    from conv5_3 down, (weight, bias) per call)."""
    from wesup_amd.engine import CONV_IDX, SIDE_OFF
    from wesup_amd.models.wesup import WESUP
    params = list(WESUP().named_parameters())
    assert sum(p.numel() for _, p in params) == 18868194
    offs, sizes, total = {}, {}, 0
    for n, p in params:
        offs[n] = total
        sizes[n] = (p.numel() + 63) // 64 * 64
        total += sizes[n]
    order = [['classifier.0.weight', 'classifier.0.bias'] + [f'fc_layers.{k}.{t}' for k in (0, 2, 4) for t in ('weight', 'bias')]]
    order += [[f'side_conv{off}.weight', f'side_conv{off}.bias'] for off in reversed(SIDE_OFF)]
    order += [[f'backbone.{i}.weight', f'backbone.{i}.bias'] for i in reversed(CONV_IDX)]
    assert sorted(n for call in order for n in call) == sorted(offs)
    return offs, sizes, total, order


def _worker8(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        offs, sizes, total, order = _model_layout()
        flat = torch.full((total,), float(rank + 1))
        red = ddp.GradAllReducer(flat, offs, sizes, bucket_bytes=16 << 20)
        for call in order:
            red.ready(call)
        launched = red.finish()
        expect = float(world * (world + 1) // 2)
        ok = bool((flat == expect).all())
        cov = sorted(launched)
        tiles = cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
        # every rank must have cut the same buckets in the same order (or the collectives would pair up wrongly)
        mine = torch.tensor([v for ab in launched for v in ab], dtype=torch.int64)
        theirs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(theirs, mine)
        same = all(torch.equal(t, mine) for t in theirs)
        if rank == 0:
            out.put((ok, tiles, same, [(b - a) * 4 for a, b in launched], total))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world8_on_the_model_layout():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, out)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, tiles, same, bucket_bytes, total = out.get(timeout=10)
    assert ok, 'all-reduced flat gradient != sum over the 8 ranks'
    assert tiles, 'buckets do not tile the flat buffer exactly once'
    assert same, 'ranks cut different buckets'
    assert total * 4 >= 18868194 * 4 and sum(bucket_bytes) == total * 4
    # On the GPU a bucket never mixes streams (head / side convs / backbone: 13.0 + 3.6 + 3 x 18.9 + 2.2 MB);
    # without streams the same reports coalesce across those borders: every bucket but the last reaches the 16 MB bar
    assert 4 <= len(bucket_bytes) <= 6 and all(b >= 16 << 20 for b in bucket_bytes[:-1]) and max(bucket_bytes) < 32 << 20


def test_non_adjacent_range_keeps_the_stream_of_its_call():
    """after an in-loop flush the new pending range must still remember the stream it was reported on (CPU:
    the bookkeeping only -- no stream exists -- but the range logic is the same)."""
    names, offs, sizes, total = _layout()
    flat = torch.zeros(total)
    red = ddp.GradAllReducer(flat, offs, sizes, bucket_bytes=1 << 30)
    red.ready(['fc.weight', 'backbone.0.weight'])          # not adjacent: the first is flushed, the second is pending
    assert red.launched == [(offs['fc.weight'], offs['fc.weight'] + sizes['fc.weight'])]
    assert (red.lo, red.hi) == (0, sizes['backbone.0.weight'])
    red.finish()


def test_shard_sampler_world8_epochs():
    n = 165                                                 # GlaS has 85 training images, CRAG 173; any odd count
    per_epoch = []
    for epoch in range(3):
        seen = []
        for r in range(8):
            s = ddp.ShardSampler(n, r, 8, seed=11)
            s.set_epoch(epoch)
            idx = list(s)
            assert len(idx) == len(s) == (n + 7) // 8
            seen.append(idx)
        flat = [i for idx in seen for i in idx]
        assert set(flat) == set(range(n)) and len(flat) == (n + 7) // 8 * 8
        per_epoch.append(seen)
    assert per_epoch[0] != per_epoch[1] != per_epoch[2]
