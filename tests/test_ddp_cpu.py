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
