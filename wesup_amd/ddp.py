"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code.  Images are independent units, so the
only exchange per iteration is the sum of the 18.87 M fp32 parameter gradients.  The engine
finishes gradients in contiguous, descending ranges of the flat gradient buffer (head layers
first, conv1_1 last); each time a range of >= ``bucket_bytes`` is complete, an asynchronous
all-reduce of that slice is queued on RCCL's stream while backward continues on the compute
stream.  ``finish()`` joins the streams before the optimiser; the 1/world_size average is folded
into the SGD kernel.  Works with any torch.distributed backend (gloo on CPU for tests)."""
import warnings

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, flat_grad, offsets, sizes, group=None, bucket_bytes=16 << 20):
        """flat_grad: 1-D tensor; offsets/sizes: dict name -> start element / padded element count."""
        self.flat = flat_grad
        self.off = offsets
        self.size = sizes
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.works = []
        self.lo = self.hi = None
        self.launched = []           # (lo, hi) ranges, for tests
        self.force = False           # also all-reduce with a single rank (exercises the RCCL path)
        self._pending_stream = None  # the stream whose work completes the pending range (GPU only)
        # measurement (bench.py): per bucket an event at its launch, an event pair around the waits of finish()
        self.profile = False
        self.t_backward = None       # event recorded by the trainer in front of backward()
        self._marks, self._finish_marks = [], None
        self._mark_history = []      # (t_backward, bucket marks, finish marks) of the profiled iterations not yet collected

    def _flush(self):
        if self.lo is None:
            return
        if self.flat.is_cuda and self._pending_stream is not None and \
                self._pending_stream != torch.cuda.current_stream(self.flat.device):
            with torch.cuda.stream(self._pending_stream):      # behind the stream that produced the range
                return self._flush()
        seg = self.flat[self.lo:self.hi]
        if self.profile and self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()              # on the stream the collective is ordered behind
            self._marks.append((ev, (self.hi - self.lo) * 4))
        if dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force):
            self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((self.lo, self.hi))
        self.lo = self.hi = None
        self._pending_stream = None

    def ready(self, names):
        """Called by the engine when the gradients of ``names`` are complete.  Backward finishes the flat buffer from
        its end towards its start, so within one call the names are taken in descending offset order: (weight, bias)
        pairs then extend the pending range downwards instead of looking non-adjacent and forcing a flush per layer
        (28 small all-reduces and ~1 ms of host time per step instead of 6 buckets).

        Streams: the engine reports a range from the stream that produced it (head and backbone weight gradients: the
        wgrad stream; side convs: the side stream), and an asynchronous collective is ordered behind the stream that is
        current when it is launched.  So a bucket never mixes streams: when a call arrives on another stream than the one
        the pending range was reported on, that range is launched first, behind ITS stream (no stream waits for another:
        a cross-stream wait here would park the side branch behind the wgrad stream's queue in the middle of backward)."""
        if self.flat.is_cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            if self.lo is not None and self._pending_stream is not None and self._pending_stream != cur:
                self._flush()
            self._pending_stream = cur
        for n in sorted(names, key=lambda k: -self.off[k]):
            lo, hi = self.off[n], self.off[n] + self.size[n]
            if self.lo is None:
                self.lo, self.hi = lo, hi
            elif hi == self.lo:
                self.lo = lo
            elif lo == self.hi:
                self.hi = hi
            else:                    # not adjacent: flush what we have, start a new range (produced on this call's stream)
                self._flush()
                self.lo, self.hi = lo, hi
                if self.flat.is_cuda:
                    self._pending_stream = cur
        if self.lo is not None and self.hi - self.lo >= self.bucket_elems:
            self._flush()

    def reset(self):
        """Forget a half-finished backward (an exception between the first ready() and finish()): wait for whatever
        was launched so that no collective is still writing into the buffer, and drop the pending range."""
        for w in self.works:
            try:
                w.wait()
            except RuntimeError as e:  # what a failed collective raises (dist.DistBackendError / gloo's RuntimeError): the
                # peer is gone, there is nothing left to wait for.  Anything else (a programming error) propagates.
                warnings.warn(f'GradAllReducer.reset: dropped a failed all-reduce ({type(e).__name__}: {e})')
        self.works = []
        self.lo = self.hi = None
        self.launched = []
        self._pending_stream = None
        self._stash_marks()

    def _stash_marks(self):
        if self._marks or self._finish_marks is not None:
            self._mark_history.append((self.t_backward, self._marks, self._finish_marks))
        self._marks, self._finish_marks, self.t_backward = [], None, None

    def finish(self):
        """Launch what is pending and make the current stream wait for every bucket (the host does not block: an NCCL
        work's wait() orders the current stream behind the collective)."""
        self._flush()
        prof = self.profile and self.flat.is_cuda
        if prof:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()              # behind the last backward kernel of the current stream
        for w in self.works:
            w.wait()
        if prof:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()              # behind the last collective: e0 -> e1 is what backward did not hide
            self._finish_marks = (e0, e1)
        self.works = []
        done, self.launched = self.launched, []
        return done

    def collect_stats(self):
        """After a device sync: the events of the profiled iterations since the last call as one dict per iteration --
        bucket sizes, their launch offsets from the start of backward (ms), the time backward took on the caller's stream
        and the exposed tail of finish() (ms): what the all-reduces added behind backward."""
        self._stash_marks()
        out = []
        for t0, marks, fin in self._mark_history:
            st = {'bucket_bytes': [b for _, b in marks],
                  'launch_offset_ms': [round(t0.elapsed_time(ev), 3) for ev, _ in marks] if t0 is not None else None}
            if fin is not None:
                st['exposed_ms'] = round(fin[0].elapsed_time(fin[1]), 4)
                if t0 is not None:
                    st['backward_ms'] = round(t0.elapsed_time(fin[0]), 3)
            out.append(st)
        self._mark_history = []
        return out


def attach(model, group=None, bucket_bytes=16 << 20):
    """Hook a WESUP model's engine to the all-reducer.  Returns the reducer."""
    model._ensure_engine()
    sizes = {n: (p.numel() + 63) // 64 * 64 for n, p in model.named_parameters()}
    red = GradAllReducer(model._flat_grad, model._offs, sizes, group, bucket_bytes)
    model.engine.on_grads_ready = red.ready
    model._reducer = red
    return red


def broadcast_parameters(model, src=0, group=None):
    model._ensure_engine()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(model._flat, src=src, group=group)


def shard_indices(n_items, rank, world, seed=0, epoch=0):
    """DistributedSampler-equivalent: a per-epoch permutation, padded to a multiple of world, strided by rank."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n_items, generator=g).tolist()
    total = (n_items + world - 1) // world * world
    perm += perm[:total - n_items]
    return perm[rank:total:world]


class ShardSampler(torch.utils.data.Sampler):
    """Per-rank view of shard_indices() for a DataLoader; ``set_epoch`` reshuffles (every rank draws the same
    permutation of the epoch and takes its own stride, so the partition changes from epoch to epoch too)."""

    def __init__(self, n_items, rank, world, seed=0):
        self.n_items, self.rank, self.world, self.seed = n_items, rank, world, seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        return iter(shard_indices(self.n_items, self.rank, self.world, self.seed, self.epoch))

    def __len__(self):
        return (self.n_items + self.world - 1) // self.world
