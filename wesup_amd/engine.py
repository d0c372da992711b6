"""Hand-scheduled forward/backward of the WESUP training step on the HIP kernels.

This is the MI355X replacement of the ATen op sequence the reference runs per
iteration:

  forward   pack image -> 13 x [conv3x3 (pre-ReLU tap) -> side 1x1 GEMM -> bilinear upsample into the
            pixel-major feature map] (+ 2x2 maxpool on 4 of them) -> superpixel scatter-mean ->
            fc_layers (3 GEMMs, fused bias+ReLU) -> classifier+softmax -> paint-back
            (models/wesup.py:263-304)
  backward  classifier -> fc_layers (TN GEMMs for dW, NT GEMMs with fused ReLU mask for dx) ->
            scatter-mean backward -> per layer: upsample backward, side-conv wgrad/dgrad -> main path
            from conv5_3 down: conv3x3 wgrad, conv3x3 dgrad with fused ReLU mask + accumulate into the
            side-branch gradient (maxpool backward where the layer was pooled)   (autograd in the reference,
            models/base.py:207)

Three HIP streams: the MFMA-bound convolution chain (forward convs, backward dgrads) runs on the caller's stream,
the side branch (weight repacking, 1x1 side GEMMs, pooling, upsampling) on a second, the conv weight gradients on a
third, joined by events; the memory-bound kernels and the wgrads then fill the tails and stalls of the matrix
kernels of the chain instead of extending the critical path.

Activations are NHWC fp32.  Parameter gradients are written straight into one flat buffer (the model's
parameters are views of a flat buffer too) so that SGD is one kernel and the data-parallel all-reduce is a
few large RCCL calls launched while the rest of backward still runs.
"""
from collections import OrderedDict

import os

import torch

from . import ops

CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
CONV_CH = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
           (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]
POOL_AFTER = [False, True, False, True, False, False, True, False, False, True, False, False, False]
SIDE_OFF = [0, 32, 64, 128, 192, 320, 448, 576, 832, 1088, 1344, 1600, 1856]
FM_CHANNELS = 2112
# (the 13th conv is followed by a MaxPool in VGG16 whose output the reference discards, models/wesup.py:279)


class KernelTimer:
    """Optional HIP-event timing of kernel classes on the launch stream (bench.py's roofline leg)."""

    def __init__(self):
        self.enabled = False
        self.only = None           # optional set of tags: every other class runs without events
        self.pending = []          # (tag, start_event, end_event, work)
        self.totals = {}
        self._free = []            # recycled events (an event record is not free: it fences the queue it is put on)
        self.replaced = 0          # event pairs with an impossible duration (see collect)
        self.replaced_pairs = []   # ... each of them as (tag, measured ms, median ms of its class): what was substituted, auditable

    def _event(self):
        return self._free.pop() if self._free else torch.cuda.Event(enable_timing=True)

    def begin(self, tag):
        if not self.enabled or (self.only is not None and tag not in self.only):
            return None
        s = self._event()
        s.record()                 # on the current stream = the stream the kernel is launched on
        return (tag, s)

    def end(self, tok, work=0.0):
        if tok is None:
            return
        e = self._event()
        e.record()
        self.pending.append((tok[0], tok[1], e, work))

    def collect(self):
        """Call after a device sync.  Returns {tag: (ms_total, launches, work_total)}.
        A pair whose end timestamp is off by tens of milliseconds turns up about once per thousand pairs on this stack (the
        next event on the same stream carries the same offset; the step itself has no such gap): a value above 5 ms AND above
        100 x the median of its class is replaced by that median, counted in self.replaced and listed in self.replaced_pairs (the
        bench line carries the list, so the substitution can be audited)."""
        by_tag = {}
        for tag, s, e, work in self.pending:
            by_tag.setdefault(tag, []).append((s.elapsed_time(e), work))
            self._free += [s, e]
        self.pending = []
        for tag, vals in by_tag.items():
            med = sorted(v for v, _ in vals)[len(vals) // 2]
            ms, n, wk = self.totals.get(tag, (0.0, 0, 0.0))
            for v, work in vals:
                if v > 5.0 and v > 100.0 * med:
                    self.replaced += 1
                    self.replaced_pairs.append((tag, round(v, 3), round(med, 4)))
                    v = med
                ms, n, wk = ms + v, n + 1, wk + work
            self.totals[tag] = (ms, n, wk)
        return self.totals

    def reset(self):
        self.pending, self.totals = [], {}


class _Bufs:
    pass


def default_route(ci, co, h, w, B):
    """The convolution algorithm of one 3x3 layer: 0 = implicit GEMM (direct form), 2 = Winograd F(2x2,3x3), 4 = Winograd
    F(4x4,3x3).  Measured per layer and pass at the three benchmark shapes (tools/wino_table.py; profiles/r03_wino_table_*.txt):
    every layer with >= 64 input channels is fastest in the F(4x4) domain (1/4 of the direct form's multiply-adds, 2.25x
    the activation bytes) at 480x480, 800x800 and 1024x1024, down to the 8x8-tile maps of conv5_x; F(2x2) (4/9 of the
    multiply-adds, 4x the bytes) is slower than F(4x4) everywhere and slower than the direct kernel at 64 input channels.
    Inside the step the two 64-channel layers (conv1_2, conv2_1: HBM-bound in the domain) gain 0.5 % at 480x480 and 2 % at
    800x800 / 1024x1024 (bench.py --winograd-min-ci 128 for the A/B); the image layer (3 channels) has no Winograd form."""
    if ci < WesupEngine.WINOGRAD_CONV_MIN_CI:
        return 0
    return WesupEngine.WINOGRAD_TILE


class WesupEngine:
    # layers whose weight gradient goes through the Winograd domain when wgrad_winograd is on AND the layer's forward did
    # not (then the forward's kept V decides): measured per layer at the bench shape (tools/wino_table.py) -- faster from
    # 128 -> 256 channels up, slower below (the transformed operands cannot be amortised by the 64/128-channel GEMMs)
    WINOGRAD_MIN_CI, WINOGRAD_MIN_CO = 128, 256
    # layers whose forward and input gradient go through the Winograd domain when conv_winograd is on: every layer
    # with >= 64 input channels (conv1_2 ... conv5_3); the image layer stays on the implicit-GEMM kernel.
    # bench.py --winograd-min-ci / --winograd-tile for the A/B.
    WINOGRAD_CONV_MIN_CI = 64
    WINOGRAD_TILE = 4                    # m of F(m x m, 3x3) for those layers: 4 (default) or 2 (round 2's routing)
    # Positions of the backward walk, measured in rounds 3 - 5 (every alternative within +-0.05 ms, HISTORY.md) and fixed: the deep
    # layers' side-conv weight gradients are queued when the chain reaches conv2_1; the shallow layers' side-branch gradients at the
    # head of the weight-gradient stream; the weight gradient of the layer above the lowest trainable one stays in front of its
    # input gradient, every other one goes behind
    _DEEP_SIDE_WGRAD_AT = 2
    _WGRAD_EARLY_LAYERS = 1

    def __init__(self, params, grads, D=32):
        """params/grads: dict name -> tensor (views of the flat parameter / gradient buffers)."""
        self.p = params
        self.g = grads
        self.D = D
        self.device = next(iter(params.values())).device
        self._bufs = OrderedDict()       # (B,H,W,Kmax) -> _Bufs, least recently used first
        # Bounds of that cache: shapes, and pixels (B*H*W summed over the cached shapes; a training buffer set is ~3-9 KB per
        # pixel).  Large shapes: the training and the validation shape (two 4 x 480^2 sets are 1.8 M pixels, two 8 x 1024^2
        # sets 16.8 M: the second evicts the first).  Small multi-scale crops (batch 1 at 0.3-0.4 x 775x522: ~50 K pixels each)
        # stay until the shape bound, so a shape that comes back finds its buffers -- and its recorded step plan -- again.
        self.max_cached_shapes = 256
        self.max_cached_pixels = 10 * 1024 * 1024
        self._last = None                # buffers of the most recent forward (feature_maps() reads these)
        self.frozen = set()              # names of parameters with requires_grad=False (set by WESUP.forward)
        self._packed = None
        self._prefetched = None
        self.ctx = None
        # ---- the seven switches; everything else about the schedule is fixed
        self.fuse_pool_bwd = True        # skip the (B,HW,2112) gradient tensor: pool-bwd fused into upsample-bwd
        self.fuse_pool_fwd = True        # skip the (B,HW,2112) feature map: scatter-mean fused with the upsample
        self.two_streams = True          # side branch and weight gradients on HIP streams of their own
        # Weight gradients of the Winograd-domain layers in the domain too (the forward's V is kept): 1/4 of the MFMA work for
        # memory-bound transform passes that run beside the dgrad chain.  False: the direct implicit-GEMM weight gradient.
        self.wgrad_winograd = True
        self.conv_winograd = True        # forward / dgrad of the layers with >= 64 input channels in the Winograd domain (False: implicit GEMM)
        # ``plain``: the reference's order of operations with one launch per pass -- the parity witness of every fused form below,
        # and what the walk falls back to layer by layer where a fused form does not apply (odd sizes, unsupported widths):
        #   * side conv in FRONT of upsample + superpixel mean with the side outputs materialised (default: behind, on one row per
        #     superpixel -- the three maps are linear, the first mixes channels, the other two pixels: they commute);
        #   * the side-branch gradient of the native-resolution layers materialised by the gather kernel and accumulated into
        #     (default: gathered by the dgrad epilogue of the layer above);
        #   * the two F(4x4) transforms of an output gradient as two launches (default: one pass, ops.winograd_dual_transform);
        #   * ReLU masks / max-pool decisions re-read from the pre-ReLU activations (default: sign bits / 3-bit codes the forward
        #     leaves, 1/16 and 1/32 of the tensor);
        #   * the max-pool backward as a launch of its own on a gradient at pooled resolution (default: the dgrad epilogue).
        self.plain = False
        self._diag_skip = set()          # TIMING-ONLY diagnostics (bench.py --diag-skip): classes of launches left out, results wrong
        self.route_fn = default_route    # (ci, co, h, w, B) -> 0 | 2 | 4, consulted per layer and shape
        self._route = None               # the 13 tile sizes of the current / most recent shape
        self._side_stream = None
        self._wgrad_stream = None
        self.timer = KernelTimer()
        self.on_grads_ready = None       # callback(names) for the data-parallel layer
        self.on_tail = None              # callback(wgrad stream, names of the last layer's parameters): see backward()
        self._rot = 0
        self.buf_generation = 0          # counts buffer sets ever created: a set's `gen` (a recorded step plan holds ITS addresses)

    # what ``plain`` switches, by the names the walk uses
    commute_side = property(lambda self: not self.plain)
    gather_side_grad = property(lambda self: not self.plain)
    dual_transform = property(lambda self: not self.plain)
    compact_masks = property(lambda self: not self.plain)
    fuse_unpool = property(lambda self: not self.plain)
    matrix_pool = True          # coarse layers: upsample + scatter-mean (and backward) as GEMMs with the interpolation-pooling matrix
    fuse_side_fwd = True        # side conv of a direct-form layer with <= 128 output channels inside its conv epilogue (where not commuted)

    # ------------------------------------------------------------------ streams
    # Ordering edges between the three streams go through the library's event pool (ops.sync_record / sync_wait: slots of a
    # fixed pool, so that a recorded step plan replays the same edges, csrc/plan.hip): fixed slots for marks that are waited
    # for later, a rotating range for "stream b goes on behind what stream a holds now".
    SLOT_W0, SLOT_W, SLOT_G, SLOT_DS, SLOT_WB, SLOT_ROT0, SLOT_ROT_N = 0, 1, 2, 15, 28, 64, 192       # G_l: 2 .. 14, ds_l: 15 .. 27

    def _edge(self, src, dst):
        """dst (torch stream) waits for everything queued so far on src."""
        k = self._rot
        self._rot = k + 1 if k + 1 < self.SLOT_ROT_N else 0
        ops.sync_record(self.SLOT_ROT0 + k, src.cuda_stream)
        ops.sync_wait(self.SLOT_ROT0 + k, dst.cuda_stream)

    def _side(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def _wg(self):
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device)
        return self._wgrad_stream

    class _OnSide:
        """Run the body on the side stream after everything queued so far on the main stream."""

        def __init__(self, eng, wait_main=True):
            self.eng, self.wait_main = eng, wait_main

        def __enter__(self):
            e = self.eng
            if not e.two_streams:
                return None
            self.prev = torch.cuda.current_stream()
            if self.wait_main:
                e._edge(self.prev, e._side())
            torch.cuda.set_stream(e._side())      # (torch.cuda.stream() as a context costs 10 us a time: ~40 times per step)
            return None

        def __exit__(self, *a):
            if self.eng.two_streams:
                torch.cuda.set_stream(self.prev)
            return False

    class _On:
        """`with torch.cuda.stream(s)` without its device bookkeeping: s becomes torch's current stream, the previous one comes back."""
        __slots__ = ('s', 'prev')

        def __init__(self, s):
            self.s = s

        def __enter__(self):
            self.prev = torch.cuda.current_stream()
            torch.cuda.set_stream(self.s)
            return self.s

        def __exit__(self, *a):
            torch.cuda.set_stream(self.prev)
            return False

    def _join_side(self):
        if self.two_streams:
            self._edge(self._side(), torch.cuda.current_stream())

    # ------------------------------------------------------------------ buffers
    BYTES_PER_PIXEL = 16 * 1024      # a training buffer set before one has been measured (9.3 GiB at 4 x 480 x 480 = 10.6 KiB per pixel)

    @staticmethod
    def _set_bytes(b):
        """Device bytes a buffer set holds right now (every tensor reachable from it, storages counted once)."""
        seen, total, todo = set(), 0, [b]
        while todo:
            o = todo.pop()
            if torch.is_tensor(o):
                st = o.untyped_storage()
                if st.data_ptr() not in seen:
                    seen.add(st.data_ptr())
                    total += st.nbytes()
            elif isinstance(o, (list, tuple)):
                todo.extend(o)
            elif isinstance(o, _Bufs):
                todo.extend(vars(o).values())
        return total

    def _bytes_per_pixel(self):
        """What a buffer set costs per pixel of its batch: the largest figure among the cached sets that have been through a training
        step (their lazily allocated parts exist), else the constant above."""
        best = 0.0
        for (B, H, W, _), b in self._bufs.items():
            if getattr(b, 'train', False) and getattr(b, 'x_in', None) is not None:
                best = max(best, self._set_bytes(b) / float(B * H * W))
        return best * 1.1 if best > 0 else float(self.BYTES_PER_PIXEL)

    def _memory_short(self, pixels):
        """The cache bounds above are counts; what ends a run is the allocator failing.  Before a set for a new shape is made: is
        there room for it -- free device memory plus what torch's caching allocator holds unused?  If not the least recently used
        sets go first (a co-resident job, a smaller GPU, or other tensors of the caller's have taken the room the bounds assume).
        The need is estimated from the sets this engine has already made (measured bytes per pixel), not from a constant."""
        try:
            free, _ = torch.cuda.mem_get_info(self.device)
            if free >= pixels * self.BYTES_PER_PIXEL:      # (the common case, one driver query; walking the sets / torch's statistics cost ms)
                return False
            need = pixels * self._bytes_per_pixel()
            if free >= need:
                return False
            idle = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        except Exception:
            return False
        return free + idle < need

    def _get_bufs(self, B, H, W, Kmax, train):
        key = (B, H, W, Kmax)
        b = self._bufs.get(key)
        if b is not None:
            self._bufs.move_to_end(key)
        else:
            # Bounded cache: training on multi-scale crops (utils/data.py: random rescale per item) or inference over
            # images of varying size meets a new shape almost every call; every entry is a full set of activation and
            # gradient buffers (~0.6 GB per 480x480 image in training), so only the most recent shapes are kept and the
            # evicted buffers go back to torch's caching allocator, which hands their blocks to the next shape.
            px = lambda k: k[0] * k[1] * k[2]
            while self._bufs and (len(self._bufs) >= max(1, self.max_cached_shapes)
                                  or sum(px(k) for k in self._bufs) + px(key) > self.max_cached_pixels
                                  or self._memory_short(px(key))):
                _, old = self._bufs.popitem(last=False)
                if old is self._last:
                    self._last = None
                if self.ctx is not None and self.ctx[0] is old:
                    self.ctx = None
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        if b is None:
            b = _Bufs()
            self.buf_generation += 1
            b.gen = self.buf_generation
            b.x0 = torch.empty(B, H, W, 4, **f32)
            b.y, b.yp, b.s, b.dims, b.yr, b.yr_wanted = [], [], [], [], [], []
            b.V = [None] * 13            # Winograd-transformed layer inputs (training forward), allocated on first use
            h, w = H, W
            for l, (ci, co) in enumerate(CONV_CH):
                b.dims.append((h, w))
                b.y.append(torch.empty(B, h, w, co, **f32))
                b.s.append(None)         # side outputs: views of the group buffers below, or allocated on first use (_side_out)
                # the ReLU'd copy the next conv (forward and wgrad) reads: the pooled tensor where the layer is pooled
                # (stored ReLU'd), a second output of the conv kernel elsewhere; the last layer has no reader
                b.yr.append(None)        # allocated on first use (forward): a Winograd-domain consumer never needs it
                b.yr_wanted.append(bool(not POOL_AFTER[l] and l < 12))
                if POOL_AFTER[l]:
                    h, w = h // 2, w // 2
                    b.yp.append(torch.empty(B, h, w, co, **f32))
                else:
                    b.yp.append(None)
            # Coarse resolutions (deep layers): upsample + scatter-mean and its backward run as GEMMs with the
            # interpolation-pooling matrix Wm of the resolution; the side outputs of the layers that share a
            # resolution sit side by side in one buffer so that one GEMM per image serves all of them.
            b.groups, b.group_of = [], [None] * 13
            if self.fuse_pool_fwd and self.fuse_pool_bwd and self.matrix_pool and Kmax % 4 == 0:
                l = 0
                while l < 13:
                    e = l
                    while e + 1 < 13 and b.dims[e + 1] == b.dims[l]:
                        e += 1
                    gh, gw = b.dims[l]
                    # ... up to 4096 cells and a matrix of at most 16 MB per image: beyond that (1024^2 with 3025 superpixels: conv5_x's
                    # 64 x 64 map under 3072 rows = 50 MB per image, 19 GF per image and direction) the gather form with the side
                    # conv commuted is faster (8 x 1024^2: 62.3 -> 61.2 ms, round 6); below it the matrix form is (batch 1: 7 - 9 %)
                    if (gh, gw) != (H, W) and gh * gw <= 4096 and (gh * gw) % 4 == 0 and Kmax * gh * gw <= (4 << 20):
                        g = _Bufs()
                        g.layers, g.h, g.w = list(range(l, e + 1)), gh, gw
                        g.off = SIDE_OFF[l]
                        g.C = sum(CONV_CH[i][1] // 2 for i in g.layers)
                        g.s = g.ds = None    # side outputs / their gradients side by side: allocated on first use
                        g.Wm = torch.empty(B, Kmax, gh * gw, **f32)
                        g.WmT = torch.empty(B, gh * gw, Kmax, **f32)
                        for i in g.layers:
                            b.group_of[i] = len(b.groups)
                        b.groups.append(g)
                    l = e + 1
            # the (B,HW,2112) feature map only exists on the unfused path (or when somebody asks for it)
            b.fm = None if self.fuse_pool_fwd else torch.empty(B, H, W, FM_CHANNELS, **f32)
            b.fm_valid = False
            b.ybar, b.dybar = [None] * 13, [None] * 13       # commuted side branch: mean_r(upsample(y_l)) and its gradient
            b.dM, b.bpart, b.dV = [None] * 13, [None] * 13, None   # dual_transform: per layer dM + bias rows, one shared V'
            b.mbits, b.pcode = [None] * 13, [None] * 13      # compact_masks: sign bits / pooling codes of y_l (None: not kept)
            b.mbits_ok, b.pcode_ok = [False] * 13, [False] * 13
            b.s_valid = [False] * 13
            b.shape = (B, H, W)
            R = B * Kmax
            b.sp_in = torch.empty(B, Kmax, FM_CHANNELS, **f32)
            b.h1 = torch.empty(R, 1024, **f32)
            b.h2 = torch.empty(R, 1024, **f32)
            b.feats = torch.empty(R, self.D, **f32)
            b.sp_pred = torch.empty(R, 2, **f32)
            b.pred = torch.empty(B, H, W, **f32)
            b.train = False
            self._bufs[key] = b
        if train and not b.train:
            R = B * Kmax
            b.G = [torch.empty_like(y) for y in b.y]
            b.ds = [None] * 13           # views of the group buffers, or allocated on first use (_side_grad)
            b.dxp = [None if yp is None else torch.empty(yp.shape[0], yp.shape[1], yp.shape[2], CONV_CH[l + 1][0], **f32)
                     for l, yp in enumerate(b.yp)]
            b.dfm = None if self.fuse_pool_bwd else torch.empty(B, H, W, FM_CHANNELS, **f32)
            b.dfeat = torch.empty(R, self.D, **f32)
            b.dh2 = torch.empty(R, 1024, **f32)
            b.dh1 = torch.empty(R, 1024, **f32)
            b.gsp = torch.empty(B, Kmax, FM_CHANNELS, **f32)
            # partial sums of the classifier's weight gradient between ops.head_bwd and ops.classifier_bwd_finish (another stream,
            # three GEMMs later): this set's own buffer, never a shared workspace a growth elsewhere could replace in between
            b.cls_part = ops.head_bwd_partials(R, self.D, dev)
            b.train = True
        return b

    def _side_out(self, b, l):
        """Buffer of layer l's side output at native resolution (B,h,w,C/2); None where the unfused path writes the side
        conv straight into the feature map (full-resolution layers)."""
        if b.s[l] is None and b.group_of[l] is not None:
            g = b.groups[b.group_of[l]]
            g.s = torch.empty(b.shape[0], g.h, g.w, g.C, dtype=torch.float32, device=self.device)
            for i in g.layers:
                c0 = SIDE_OFF[i] - g.off
                b.s[i] = g.s[..., c0:c0 + CONV_CH[i][1] // 2]
        elif b.s[l] is None and not (b.dims[l] == b.shape[1:] and not self.fuse_pool_fwd):
            h, w = b.dims[l]
            b.s[l] = torch.empty(b.shape[0], h, w, CONV_CH[l][1] // 2, dtype=torch.float32, device=self.device)
        return b.s[l]

    def _side_grad(self, b, l):
        if b.ds[l] is None and b.group_of[l] is not None:
            g = b.groups[b.group_of[l]]
            g.ds = torch.empty(b.shape[0], g.h, g.w, g.C, dtype=torch.float32, device=self.device)
            for i in g.layers:
                c0 = SIDE_OFF[i] - g.off
                b.ds[i] = g.ds[..., c0:c0 + CONV_CH[i][1] // 2]
        elif b.ds[l] is None and not (b.dims[l] == b.shape[1:] and not self.fuse_pool_bwd):
            h, w = b.dims[l]
            b.ds[l] = torch.empty(b.shape[0], h, w, CONV_CH[l][1] // 2, dtype=torch.float32, device=self.device)
        return b.ds[l]

    def _commuted(self, b, l):
        """Layer l's side conv behind the pooling instead of in front of it (see commute_side)."""
        return (self.commute_side and self.fuse_pool_fwd and self.fuse_pool_bwd
                and b.group_of[l] is None)

    def bufs_gen(self, B, H, W, Kmax):
        """The identity of the cached buffer set of a shape (None: not cached), marking it most recently used: what a
        recorded step plan of that shape depends on -- other shapes' sets coming and going do not move its addresses."""
        b = self._bufs.get((B, H, W, Kmax))
        if b is None:
            return None
        self._bufs.move_to_end((B, H, W, Kmax))
        return b.gen

    def release_buffers(self):
        self._bufs.clear()
        self._last = None
        self.ctx = None

    # ------------------------------------------------------------------ weights
    def route(self, B, H, W):
        """Per layer: m of the Winograd domain its forward / input gradient (and, with the kept V, weight gradient) run in,
        or 0 for the implicit-GEMM kernel."""
        r, h, w = [], H, W
        for l, (ci, co) in enumerate(CONV_CH):
            r.append(int(self.route_fn(ci, co, h, w, B)) if (self.conv_winograd and ci >= 32) else 0)
            if POOL_AFTER[l]:
                h, w = h // 2, w // 2
        return r

    def _pack_weights(self, train):
        pk = self._packed
        if pk is None:
            pk = _Bufs()
            pk.wf, pk.wd = [], []
            pk.uf, pk.ud = [None] * 13, [None] * 13          # Winograd-domain filters, allocated on first use
            pk.wino = None
            pk.bwd_todo, pk.bwd_ready = None, False
            for l, (ci, co) in enumerate(CONV_CH):
                pk.wf.append(torch.empty(co, ops.conv3x3_kpad(ci), dtype=torch.float32, device=self.device))
                pk.wd.append(None if l == 0 else torch.empty(ci, 9 * co, dtype=torch.float32, device=self.device))
            # (one flat buffer: the transposed side weights of equally wide layers sit at a constant stride -- batched launches)
            flatT = torch.empty(sum(co * (co // 2) for ci, co in CONV_CH), dtype=torch.float32, device=self.device)
            pk.sideT, o = [], 0
            for ci, co in CONV_CH:
                pk.sideT.append(flatT[o:o + co * (co // 2)].view(co, co // 2))
                o += co * (co // 2)
            pk.fcT = [torch.empty(FM_CHANNELS, 1024, dtype=torch.float32, device=self.device),
                      torch.empty(1024, 1024, dtype=torch.float32, device=self.device),
                      torch.empty(1024, self.D, dtype=torch.float32, device=self.device)]
            self._packed = pk
        wino = list(self._route) if self._route is not None else self.route(4, 480, 480)
        if self._prefetched == train and pk.wino == wino:     # prefetch_weights() already queued exactly this for the current step
            self._prefetched = None
            return pk
        self._prefetched = None
        pk.wino = wino
        # ~40 launch-latency-bound repack kernels go to the side stream (idle at this point) and are joined in front of
        # the first convolution; the trainer queues them before the superpixel preprocessing (prefetch_weights)
        # conv1_1's own panel (a 5 us kernel) on the caller's stream, in front of the first convolution: the side stream starts
        # the step with the label / mask copies of the step runner, and the chain would wait for them with it
        first_here = not wino[0]
        if first_here:
            ops.pack_conv3x3_weight(self.p[f'backbone.{CONV_IDX[0]}.weight'], pk.wf[0], None, need_dgrad=False)
        with self._OnSide(self):
            pk.ready0 = pk.ready = pk.bwd_ready = False
            fwd4, dg4 = [], []                     # F(4x4) layers: one launch for the forward filters, one for the rotated ones
            for l, idx in enumerate(CONV_IDX):
                if l == 0 and first_here:
                    continue
                m = wino[l]
                w = self.p[f'backbone.{idx}.weight']
                if m:
                    P = ops.winograd_positions(m)
                    if pk.uf[l] is None or pk.uf[l].shape[0] != P:
                        ci, co = CONV_CH[l]
                        pk.uf[l] = torch.empty(P, co, ci, dtype=torch.float32, device=self.device)
                        pk.ud[l] = torch.empty(P, ci, co, dtype=torch.float32, device=self.device)
                    if m == 4:
                        fwd4.append((w, pk.uf[l], None))
                        dg4.append((w, None, pk.ud[l]))
                    else:
                        ops.winograd_pack_weight(w, need_dgrad=False, u_fwd=pk.uf[l], m=m)
                    continue
                ops.pack_conv3x3_weight(w, pk.wf[l], None, need_dgrad=False)
                if l == 0 and self.two_streams:
                    ops.sync_record(self.SLOT_W0)
                    pk.ready0 = True
            if fwd4:
                ops.winograd_pack_weights(fwd4)
            if self.two_streams:
                ops.sync_record(self.SLOT_W)
                pk.ready = True
            # What only the backward reads -- the rotated F(4x4) filters of the input gradients, the dgrad panels of the direct
            # layers, the transposed side / fc weights -- is not queued here: at the head of the step it shared the memory system
            # with conv1_1 and conv1_2's input transform (0.4 GB written beside kernels the chain waits for).  forward() queues it
            # behind its last pooling, where it runs beside the fc layers and the loss section (round 5).
            pk.bwd_todo = (dg4, wino) if train else None
        return pk

    def _pack_weights_bwd(self, pk):
        """The backward's share of the weight repacking (see _pack_weights), on the side stream; SLOT_WB marks its end."""
        todo, pk.bwd_todo = pk.bwd_todo, None
        if todo is None:
            return
        dg4, wino = todo
        with self._OnSide(self):
            if dg4:
                ops.winograd_pack_weights(dg4)
            for l, idx in enumerate(CONV_IDX):
                if wino[l] == 4:
                    continue
                if wino[l]:
                    ops.winograd_pack_weight(self.p[f'backbone.{idx}.weight'], need_fwd=False, u_dgrad=pk.ud[l], m=wino[l])
                elif l > 0:
                    ops.pack_conv3x3_weight(self.p[f'backbone.{idx}.weight'], None, pk.wd[l], need_fwd=False)
            # the panels the input-gradient GEMMs of the side convs / fc layers read: 16 transposes, one launch
            tr = [(self.p[f'side_conv{off}.weight'].view(CONV_CH[l][1] // 2, CONV_CH[l][1]), pk.sideT[l])
                  for l, off in enumerate(SIDE_OFF)]
            tr += [(self.p[f'fc_layers.{k}.weight'], pk.fcT[i]) for i, k in enumerate((0, 2, 4))]
            ops.transpose_batched(tr)
            if self.two_streams:
                ops.sync_record(self.SLOT_WB)
                pk.bwd_ready = True
        return pk

    def _wino(self, l):
        """m of the Winograd domain layer l runs in at the current shape (0: implicit GEMM)."""
        return self._route[l]

    def prefetch_weights(self, train=True):
        """Queue the weight repacking of the coming forward now (side stream, behind everything queued so far, i.e.
        behind the optimiser step).  The parameters must not change between this call and the forward."""
        self._prefetched = None
        self._pack_weights(train)
        self._prefetched = train

    def side_stream(self):
        """Context: run the body on the side stream behind everything queued so far on the current stream.  The trainer
        puts the superpixel preprocessing there: the conv chain does not need it, only the pooling (side stream) and
        the kernels behind the forward join do."""
        return self._OnSide(self)

    # ------------------------------------------------------------------ forward
    def forward(self, img, meta, train=True, need_paint=True, head=True):
        """img (B,3,H,W) fp32 on the GPU, meta = ops.sp_preprocess(...).  Returns (feats, sp_pred, pred)
        shaped (B,Kmax,D), (B,Kmax,2), (B,H,W); buffers are reused by the next call of the same shape."""
        B, _, H, W = img.shape
        Kmax = meta.Kmax
        assert (meta.B, meta.H, meta.W) == (B, H, W)
        if torch.cuda.current_device() != self.device.index:     # every launch goes to the CURRENT device's current stream
            raise RuntimeError(f'the model lives on {self.device} but the current device is cuda:{torch.cuda.current_device()}: '
                               'one process per GPU (torch.cuda.set_device) -- launches would go to the wrong device')
        b = self._get_bufs(B, H, W, Kmax, train)
        self._last = b
        self._route = self.route(B, H, W)
        pk = self._pack_weights(train)
        p = self.p
        T = self.timer
        ops.pack_input(img, b.x0)
        if b.groups:
            with self._OnSide(self):     # the side stream is idle until conv1_1 is done
                tok = T.begin('interp_matrix')
                for g in b.groups:
                    ops.sp_interp_matrix(meta, g.h, g.w, out=g.Wm)
                ops.transpose_batched([(g.Wm[i], g.WmT[i]) for g in b.groups for i in range(B)])
                T.end(tok, 0.0)
        fused = self.fuse_pool_fwd
        pending_side = None              # the side-branch work of the previous layer, when it is queued behind this layer's transform
        cur, cur_relu = b.x0, False      # the layer's input tensor, and whether its ReLU is still to be applied on load
        b.x_in, b.x_relu = [None] * 13, [False] * 13
        b.wino_fwd = list(self._route)
        b.relu_stored = all(b.yr_wanted[l] for l in range(12) if not POOL_AFTER[l])
        b.fm_valid = not fused
        fm2d = None if fused else b.fm.view(B * H * W, FM_CHANNELS)
        for l, (ci, co) in enumerate(CONV_CH):
            h, w = b.dims[l]
            idx, off = CONV_IDX[l], SIDE_OFF[l]
            if l == 0 and pk.ready0:
                ops.sync_wait(self.SLOT_W0)
            if l == 1 and pk.ready:
                ops.sync_wait(self.SLOT_W)
            ws = p[f'side_conv{off}.weight'].view(co // 2, co)
            commute = self._commuted(b, l)
            b.s_valid[l] = not commute
            # unfused, full resolution: the side conv writes its slice of fm directly
            s_l = None if commute else self._side_out(b, l)
            s2d = None if commute else (fm2d[:, off:off + co // 2] if s_l is None else s_l.view(B * h * w, co // 2))
            # optional: the side conv of the four widest layers (64 / 128 channels at 480^2 / 240^2: the y re-read is
            # 236 / 118 MB) in the conv's epilogue, where the output tile sits in LDS anyway
            side_in_conv = self.fuse_side_fwd and co <= 128 and not self._wino(l) and not commute
            b.x_in[l], b.x_relu[l] = cur, cur_relu
            # The ReLU'd copy of this layer's output exists for the next layer's 9-tap re-reads and its weight gradient.
            # A Winograd-domain consumer reads its input once (input transform) and its weight gradient reads the kept V:
            # then the copy is not written at all and the transform applies the ReLU while loading y.
            yr = None
            if b.yr_wanted[l] and not (l < 12 and self._wino(l + 1) and (self.wgrad_winograd or not train)):
                if b.yr[l] is None:
                    b.yr[l] = torch.empty(B, h, w, co, dtype=torch.float32, device=self.device)
                yr = b.yr[l]
            m = self._wino(l)
            bits_out = code_out = None
            if l >= 1:
                b.mbits_ok[l - 1] = False
            b.pcode_ok[l] = False
            if train and self.compact_masks and m == 4:
                # sign bits of y_{l-1}: this layer's input transform reads it (pre-ReLU, not pooled) and this layer's input
                # gradient is the consumer (one-kernel route: product co -> ci)
                tiles = ops.winograd_tiles(B, h, w, 4)
                if l >= 1 and cur is b.y[l - 1] and cur_relu and ops.winograd_fused_supported(co, ci, 4, tiles) == 2:
                    if b.mbits[l - 1] is None:
                        b.mbits[l - 1] = torch.empty(B, h, w, ci // 4, dtype=torch.uint8, device=self.device)
                    bits_out = b.mbits[l - 1]
                    b.mbits_ok[l - 1] = True
                # pooling codes of y_l: this layer's pooling epilogue writes them, the input gradient of layer l + 1 (through
                # the max-pool backward, one-kernel route) reads them
                if (POOL_AFTER[l] and l < 12 and yr is None and self.fuse_unpool and self._wino(l + 1) == 4
                        and ops.winograd_fused_supported(ci, co, 4, tiles) >= 1
                        and ops.winograd_fused_supported(CONV_CH[l + 1][1], CONV_CH[l + 1][0], 4,
                                                         ops.winograd_tiles(B, h // 2, w // 2, 4)) == 2):
                    if b.pcode[l] is None:
                        b.pcode[l] = torch.empty(B, h // 2, w // 2, co // 4, dtype=torch.int16, device=self.device)
                    code_out = b.pcode[l]
                    b.pcode_ok[l] = True
            if m:
                vshape = (ops.winograd_positions(m), ops.winograd_tiles(B, h, w, m), ci)
                if train and (b.V[l] is None or b.V[l].shape != vshape):      # the transformed input, kept for the weight gradient
                    b.V[l] = torch.empty(vshape, dtype=torch.float32, device=self.device)
                # timed as 'winograd_gemm' (executed MFMA FLOPs: 4/9 resp. 1/4 of the direct form's) + 'winograd_transform'
                # (bytes).  (An m x m output tile holds whole windows of the max-pool behind conv2_2 / conv3_3 / conv4_3: the
                # output transform writes the pooled tensor too and the max-pool launch below is skipped)
                after, pending_side = pending_side, None
                ops.conv3x3_fwd_winograd(cur, pk.uf[l], p[f'backbone.{idx}.bias'], relu_in=cur_relu, out=b.y[l],
                                         out_relu=yr, v_keep=b.V[l] if train else None, ws_tag='wino_main', timer=T,
                                         out_pool=b.yp[l] if POOL_AFTER[l] else None, pool_relu=b.relu_stored, m=m,
                                         relu_bits_out=bits_out, pool_code_out=code_out, after_transform=after)
            else:
                if pending_side is not None:
                    pending_side()
                    pending_side = None
                tok = T.begin('conv3x3_fwd')
                ops.conv3x3_fwd(cur, pk.wf[l], p[f'backbone.{idx}.bias'], co, relu_in=cur_relu, out=b.y[l], out_relu=yr,
                                side=(ws, p[f'side_conv{off}.bias'], s2d) if side_in_conv else None)
                T.end(tok, 2.0 * B * h * w * co * ((3 if l == 0 else ci) * 9 + (co // 2 if side_in_conv else 0)))
            # side branch of this layer: 1x1 conv on the pre-ReLU tap, then either the fused upsample+scatter-mean
            # straight into the superpixel feature slice, or upsample into fm's channel slice
            def side_work(l=l, ci=ci, co=co, h=h, w=w, off=off, ws=ws, commute=commute, s_l=s_l, s2d=s2d, side_in_conv=side_in_conv):
              with self._OnSide(self):
                grp = b.groups[b.group_of[l]] if b.group_of[l] is not None else None
                if ('side_fwd_shallow' in self._diag_skip and grp is None) or ('side_fwd_deep' in self._diag_skip and grp is not None):
                    pass                     # timing-only diagnostic: sp_in keeps an earlier step's slice
                elif commute:
                    if b.ybar[l] is None:
                        b.ybar[l] = torch.empty(B, Kmax, co, dtype=torch.float32, device=self.device)
                    tok = T.begin('sp_pool_up_fwd')          # (commuted layers are the gather layers: no interpolation matrix)
                    ops.sp_pool_upsample_fwd(b.y[l], meta, b.ybar[l], 0)
                    T.end(tok, 4.0 * B * (h * w * co + H * W + Kmax * co))
                    tok = T.begin('side_fwd')
                    ops.gemm_nt(b.ybar[l].view(B * Kmax, co), ws, p[f'side_conv{off}.bias'],
                                out=b.sp_in.view(B * Kmax, FM_CHANNELS)[:, off:off + co // 2])
                    T.end(tok, 2.0 * B * Kmax * co * (co // 2))
                elif not side_in_conv:
                    tok = T.begin('side_fwd')
                    ops.gemm_nt(b.y[l].view(B * h * w, co), ws, p[f'side_conv{off}.bias'], out=s2d)
                    T.end(tok, 2.0 * B * h * w * co * (co // 2))
                if commute or ('side_fwd_deep' in self._diag_skip and grp is not None):
                    pass
                elif b.group_of[l] is not None:
                    g = b.groups[b.group_of[l]]
                    if l == g.layers[-1]:        # all side outputs of this resolution are in: sp_in slice = Wm . s
                        tok = T.begin('sp_pool_mat_fwd')
                        ops.gemm_tn_batched(g.WmT, g.s.view(B, g.h * g.w, g.C), b.sp_in[:, :, g.off:g.off + g.C],
                                            ws_tag='side')
                        T.end(tok, 2.0 * B * Kmax * g.h * g.w * g.C)
                elif fused:
                    tok = T.begin('sp_pool_up_fwd')
                    ops.sp_pool_upsample_fwd(s_l, meta, b.sp_in, off)
                    T.end(tok, 4.0 * B * (h * w * (co // 2) + H * W + Kmax * (co // 2)))
                elif s_l is not None:
                    tok = T.begin('upsample_fwd')
                    ops.upsample_fwd(s_l, b.fm, off)
                    T.end(tok, 4.0 * B * H * W * (co // 2))
            # Queued now, the memory-bound pooling of y_l would run beside the equally memory-bound input transform of layer
            # l + 1 (both read y_l); deferred until that transform has been queued (ops: after_transform), it runs beside the
            # layer's products instead -- a memory-bound kernel next to an MFMA-bound one.
            # (the deep layers too: deferring only the memory-bound shallow ones measured 0.03 ms worse)
            if self.two_streams and l < 12 and self._wino(l + 1):
                pending_side = side_work
            else:
                side_work()
            if POOL_AFTER[l]:
                if not self._wino(l):
                    ops.maxpool2_fwd(b.y[l], b.yp[l], relu=b.relu_stored)
                cur, cur_relu = b.yp[l], not b.relu_stored
            elif yr is not None:
                cur, cur_relu = yr, False
            else:
                cur, cur_relu = b.y[l], True
        self._join_side()
        if train:
            self._pack_weights_bwd(pk)
        if not fused:
            tok = T.begin('sp_pool_fwd')
            ops.sp_pool_fwd(b.fm, meta, out=b.sp_in)
            T.end(tok, 4.0 * B * (FM_CHANNELS * H * W + H * W + Kmax * FM_CHANNELS))
        R = B * Kmax
        tok = T.begin('mlp_fwd')
        ops.gemm_nt(b.sp_in.view(R, FM_CHANNELS), p['fc_layers.0.weight'], p['fc_layers.0.bias'], out=b.h1, flags=ops.RELU_OUT)
        ops.gemm_nt(b.h1, p['fc_layers.2.weight'], p['fc_layers.2.bias'], out=b.h2, flags=ops.RELU_OUT)
        ops.gemm_nt(b.h2, p['fc_layers.4.weight'], p['fc_layers.4.bias'], out=b.feats, flags=ops.RELU_OUT)
        T.end(tok, 2.0 * R * (FM_CHANNELS * 1024 + 1024 * 1024 + 1024 * self.D))
        # head=False: the caller runs the classifier together with the label propagation (ops.head_fwd, the step runner) and
        # paints behind it; sp_pred is then not yet filled when this returns
        if head:
            ops.classifier_fwd(b.feats, p['classifier.0.weight'], p['classifier.0.bias'], b.sp_pred)
        sp_pred3 = b.sp_pred.view(B, Kmax, 2)
        if need_paint and head:
            ops.paint_fwd(sp_pred3, meta, 1, out=b.pred)
        self.ctx = (b, pk, meta, B, H, W, Kmax) if train else None
        return b.feats.view(B, Kmax, self.D), sp_pred3, b.pred

    def feature_maps(self):
        """(B,H,W,2112) pixel-major feature maps of the last forward (models/wesup.py:280).  On the fused path they
        are never needed by the step itself; they are materialised here on demand from the saved side outputs."""
        b = self._last
        if b is None:
            return None
        if not b.fm_valid:
            B, H, W = b.shape
            if b.fm is None:
                b.fm = torch.empty(B, H, W, FM_CHANNELS, dtype=torch.float32, device=self.device)
            for l, off in enumerate(SIDE_OFF):
                s_l = self._side_out(b, l)
                if not b.s_valid[l]:          # commuted side branch: the step never formed this side output
                    h, w = b.dims[l]
                    co = CONV_CH[l][1]
                    ops.gemm_nt(b.y[l].view(B * h * w, co), self.p[f'side_conv{off}.weight'].view(co // 2, co),
                                self.p[f'side_conv{off}.bias'], out=s_l.view(B * h * w, co // 2))
                    b.s_valid[l] = True
                ops.upsample_fwd(s_l.contiguous(), b.fm, off)
            b.fm_valid = True
        return b.fm

    # ------------------------------------------------------------------ backward
    def abort_backward(self):
        """A backward that raised half-way (the step runner's NaN check fires inside it): the caller's stream waits for the side and
        weight-gradient streams, the context of the forward is dropped.  The gradient buffers hold garbage afterwards."""
        if self.two_streams:
            main = torch.cuda.current_stream()
            if self._wgrad_stream is not None:
                self._edge(self._wgrad_stream, main)
            if self._side_stream is not None:
                self._edge(self._side_stream, main)
        self.ctx = None

    def backward(self, dfeat_extra, dpred, head_done=False):
        """dpred (B,Kmax,2) [and optional dfeat_extra (B,Kmax,D)]: gradients of the loss w.r.t. sp_pred /
        sp_features.  Writes every parameter gradient into self.g (overwrites)."""
        assert self.ctx is not None, 'backward without a training-mode forward'
        b, pk, meta, B, H, W, Kmax = self.ctx
        p, g, T = self.p, self.g, self.timer
        R = B * Kmax
        D = self.D
        ready = self.on_grads_ready or (lambda names: None)
        # Frozen backbone layers (freeze_backbone, models/wesup.py:427-429: requires_grad=False + the optimiser's filter):
        # autograd in the reference then never computes their weight gradients nor any activation gradient that only
        # they would need.  Same here: a frozen layer has no wgrad launch, and below the lowest trainable layer there
        # is no dgrad chain and no side-conv dgrad at all (the side convs' own wgrads only need ds_l and y_l).
        trainable = [not {f'backbone.{i}.weight', f'backbone.{i}.bias'} <= self.frozen for i in CONV_IDX]
        lowest = min([l for l in range(13) if trainable[l]], default=13)
        # ---- classifier + fc_layers.  Between the last forward conv and the first dgrad the step has ONE chain of small
        # kernels (head forward, loss, head backward, pooling backward of the deepest layers: ~1.7 ms in which the chip
        # is mostly idle), so only what that chain needs stays on it: the input-gradient GEMMs dfeat -> dh2 -> dh1 -> gsp.
        # The three weight-gradient GEMMs of the fc layers (the largest of the head: 2304 x 2112 x 1024) produce
        # parameter gradients only and go to the wgrad stream, where they run beside the chain.
        if getattr(pk, 'bwd_ready', False):
            ops.sync_wait(self.SLOT_WB)
        head_names = ['classifier.0.weight', 'classifier.0.bias'] + [f'fc_layers.{k}.{t}' for k in (0, 2, 4) for t in ('weight', 'bias')]
        tok = T.begin('mlp_bwd')
        # head_done: ops.head_bwd (loss + its gradient + the classifier's backward in one launch, the step runner) has written
        # b.dfeat and left the partial sums of the classifier's weight gradient in its workspace; they are added up off the chain
        if not head_done:
            ops.classifier_bwd(b.feats, p['classifier.0.weight'], b.sp_pred, dpred.reshape(R, 2),
                               None if dfeat_extra is None else dfeat_extra.reshape(R, D),
                               b.dfeat, g['classifier.0.weight'], g['classifier.0.bias'])
        else:
            assert dfeat_extra is None
        gsp2d = b.gsp.view(R, FM_CHANNELS)
        off_chain = self.two_streams
        if head_done and not off_chain:
            ops.classifier_bwd_finish(b.cls_part, R, D, g['classifier.0.weight'], g['classifier.0.bias'])
        if not off_chain:
            ops.gemm_tn(b.dfeat, b.h2, out=g['fc_layers.4.weight'], colsum=g['fc_layers.4.bias'])
        ops.gemm_nt(b.dfeat, pk.fcT[2], None, out=b.dh2, mask=b.h2)
        if not off_chain:
            ops.gemm_tn(b.dh2, b.h1, out=g['fc_layers.2.weight'], colsum=g['fc_layers.2.bias'])
        ops.gemm_nt(b.dh2, pk.fcT[1], None, out=b.dh1, mask=b.h1)
        if not off_chain:
            ops.gemm_tn(b.dh1, b.sp_in.view(R, FM_CHANNELS), out=g['fc_layers.0.weight'], colsum=g['fc_layers.0.bias'])
        ops.gemm_nt(b.dh1, pk.fcT[0], None, out=gsp2d)
        T.end(tok, (4.0 if not off_chain else 2.0) * R * (FM_CHANNELS * 1024 + 1024 * 1024 + 1024 * D))
        if off_chain:
            wgs = self._wg()
            self._edge(torch.cuda.current_stream(), wgs)
            with self._On(wgs):
                if head_done:
                    ops.classifier_bwd_finish(b.cls_part, R, D, g['classifier.0.weight'], g['classifier.0.bias'])
                tok = T.begin('mlp_wgrad')
                ops.gemm_tn(b.dfeat, b.h2, out=g['fc_layers.4.weight'], colsum=g['fc_layers.4.bias'], ws_tag='wgrad')
                ops.gemm_tn(b.dh2, b.h1, out=g['fc_layers.2.weight'], colsum=g['fc_layers.2.bias'], ws_tag='wgrad')
                ops.gemm_tn(b.dh1, b.sp_in.view(R, FM_CHANNELS), out=g['fc_layers.0.weight'], colsum=g['fc_layers.0.bias'],
                            ws_tag='wgrad')
                T.end(tok, 2.0 * R * (FM_CHANNELS * 1024 + 1024 * 1024 + 1024 * D))
                ready(head_names)
        else:
            ready(head_names)
        # ---- scatter-mean backward (materialised) or fused into the upsample backward
        if not self.fuse_pool_bwd:
            tok = T.begin('sp_pool_bwd')
            ops.sp_pool_bwd(b.gsp, meta, out=b.dfm)
            T.end(tok, 4.0 * B * (FM_CHANNELS * H * W + H * W + Kmax * FM_CHANNELS))
            dfm2d = b.dfm.view(B * H * W, FM_CHANNELS)
        # ---- side branches (side stream), deepest layer first: ds_l, side-conv wgrad, side-conv dgrad -> G_l.
        # g_ready[l] marks "G_l holds the side-branch gradient"; the main chain accumulates into it afterwards.
        g_ready = [None] * 13
        # The gather-style upsample backward of the shallow layers only needs gsp.  Queued in layer order on the side
        # stream it sat between the side GEMMs of layers 7..1 and the dgrad chain waited for it (~0.75 ms with no MFMA
        # kernel in flight); on a stream of its own it runs under the deep layers' GEMMs.
        ds_ready = [None] * 13

        def commuted_G(ls):
            """Commuted side branch of the layers ls (one resolution, deepest first): dYbar_l = g_slice . W_side (B*Kmax rows),
            then G_l = the upsample + scatter-mean backward of dYbar_l, written straight into the conv's gradient buffer (all C
            channels) -- the layers of a coarse resolution in one launch (the scan of a cell's pixel window is shared)."""
            h, w = b.dims[ls[0]]
            for l in ls:
                co, off = CONV_CH[l][1], SIDE_OFF[l]
                if b.dybar[l] is None:
                    b.dybar[l] = torch.empty(B, Kmax, co, dtype=torch.float32, device=self.device)
                tok = T.begin('side_bwd')
                ops.gemm_nt(gsp2d[:, off:off + co // 2], pk.sideT[l], None, out=b.dybar[l].view(R, co))
                if gat[l]:      # gathered per pixel by the epilogue of layer l + 1's input gradient: rows divided by their areas here, once
                    ops.scale_rows_by_area(b.dybar[l], meta.area_new)
                T.end(tok, 2.0 * R * co * (co // 2))
            tok = T.begin('upsample_bwd')
            if (h, w) == (H, W) or sum(CONV_CH[l][1] for l in ls) > 768:
                for l in ls:
                    if not gat[l]:           # (gat: the dgrad epilogue of layer l + 1 gathers from dYbar_l itself)
                        ops.upsample_bwd_fused(b.dybar[l], meta.new_row, meta.area_new, H, W, 0, h, w, CONV_CH[l][1], out=b.G[l])
            else:
                ops.upsample_bwd_fused_group([b.dybar[l] for l in ls], meta.new_row, meta.area_new, H, W, h, w,
                                             [b.G[l] for l in ls])
            T.end(tok, 4.0 * B * sum(h * w * CONV_CH[l][1] + H * W + Kmax * CONV_CH[l][1] for l in ls if not gat[l]))

        # native-resolution commuted layers whose G is written by the dgrad epilogue of the layer above (gather form)
        gat = [False] * 13
        if self.gather_side_grad:
            for l in range(0, 12):
                ci1, co1 = CONV_CH[l + 1]
                if (b.group_of[l] is None and self._commuted(b, l) and b.dims[l] == (H, W) and l >= lowest
                        and b.wino_fwd[l + 1] == 4
                        and ops.winograd_fused_supported(co1, ci1, 4, ops.winograd_tiles(B, *b.dims[l + 1], 4)) == 2
                        and (not POOL_AFTER[l] or (self.fuse_unpool and H % 2 == 0 and W % 2 == 0))):
                    gat[l] = True

        def commuted_runs():
            """The commuted layers whose G the dgrad chain needs, deepest first, in runs of (at most three) layers that share a
            resolution."""
            runs = []
            for l in range(12, -1, -1):
                if b.group_of[l] is None and self._commuted(b, l) and l >= lowest:
                    if runs and b.dims[runs[-1][0]] == b.dims[l] and len(runs[-1]) < 3:
                        runs[-1].append(l)
                    else:
                        runs.append([l])
            return runs

        def queue_shallow():
            # ... at the head of the wgrad stream, which has nothing to do until the first weight gradient is queued.
            # (A fourth stream of its own measured the same; with three engine streams + the RCCL stream the process
            # stays within the 4 hardware queues HIP maps streams onto by default -- a fifth stream aliases two of
            # them and costs 1.2 ms per step, which is what a live process group did to the 4-stream schedule.)
            aux = self._wg()
            self._edge(torch.cuda.current_stream(), aux)
            with self._On(aux):
                for ls in commuted_runs():
                    commuted_G(ls)
                    ops.sync_record(self.SLOT_G + ls[0])
                    for l in ls:
                        g_ready[l] = self.SLOT_G + ls[0]
                for l in range(12, -1, -1):
                    if b.group_of[l] is None and self._commuted(b, l):
                        pass
                    elif b.group_of[l] is None:
                        h, w = b.dims[l]
                        tok = T.begin('upsample_bwd')
                        ops.upsample_bwd_fused(b.gsp, meta.new_row, meta.area_new, H, W, SIDE_OFF[l], h, w,
                                               CONV_CH[l][1] // 2, out=self._side_grad(b, l))
                        # own byte model (pool-backward fused in): the gradient at native resolution out, the pixel labels
                        # and one row of g per superpixel in -- not the (H, W, C/2) slice of a materialised gradient
                        T.end(tok, 4.0 * B * (h * w * (CONV_CH[l][1] // 2) + H * W + Kmax * (CONV_CH[l][1] // 2)))
                        ds_ready[l] = self.SLOT_DS + l
                        ops.sync_record(ds_ready[l])

        # the shallow layers' side-branch gradients: at the head of the weight-gradient stream
        if self.two_streams and self.fuse_pool_bwd:
            queue_shallow()
        ds2ds = [None] * 13

        def side_wgrad(l):
            if 'side_wgrad' in self._diag_skip:
                return
            co = CONV_CH[l][1]
            h, w = b.dims[l]
            off = SIDE_OFF[l]
            P = B * h * w
            tok = T.begin('side_bwd')
            if self._commuted(b, l):         # dW = g_slice^T . Ybar, db = column sums of g_slice (the rows of upsample+mean sum to 1)
                ops.gemm_tn(gsp2d[:, off:off + co // 2], b.ybar[l].view(R, co), out=g[f'side_conv{off}.weight'].view(co // 2, co),
                            ws_tag='side', colsum=g[f'side_conv{off}.bias'])
                T.end(tok, 2.0 * R * co * (co // 2))
            else:
                ops.gemm_tn(ds2ds[l], b.y[l].view(P, co), out=g[f'side_conv{off}.weight'].view(co // 2, co), ws_tag='side',
                            colsum=g[f'side_conv{off}.bias'])
                T.end(tok, 2.0 * P * co * (co // 2))
            # reported from the side stream, layer by layer (the reducer orders a bucket behind every stream that
            # contributed to it): the side-conv gradients leave with the head's bucket instead of after the final join
            ready([f'side_conv{off}.weight', f'side_conv{off}.bias'])

        with self._OnSide(self):
            for l in range(12, -1, -1):
                ci, co = CONV_CH[l]
                h, w = b.dims[l]
                off = SIDE_OFF[l]
                P = B * h * w
                if self._commuted(b, l):
                    if not self.two_streams and l >= lowest:     # single-stream schedule: not queued above
                        for ls in commuted_runs():
                            if ls[0] == l:
                                commuted_G(ls)
                    continue
                self._side_grad(b, l)            # (allocated on first use; a group's buffer serves all its layers)
                if b.group_of[l] is not None:
                    grp = b.groups[b.group_of[l]]
                    if l == grp.layers[-1]:      # ds of every layer of this resolution at once: ds = Wm^T . gsp slice
                        tok = T.begin('upsample_mat_bwd')
                        ops.gemm_tn_batched(grp.Wm, b.gsp[:, :, grp.off:grp.off + grp.C],
                                            grp.ds.view(B, grp.h * grp.w, grp.C), ws_tag='side')
                        T.end(tok, 2.0 * B * Kmax * grp.h * grp.w * grp.C)
                    tok = None
                    ds2d = b.ds[l].view(P, co // 2)
                elif ds_ready[l] is not None:
                    ops.sync_wait(ds_ready[l])
                    tok = None
                    ds2d = b.ds[l].view(P, co // 2)
                elif self.fuse_pool_bwd:
                    tok = T.begin('upsample_bwd')
                    ops.upsample_bwd_fused(b.gsp, meta.new_row, meta.area_new, H, W, off, h, w, co // 2, out=self._side_grad(b, l))
                    ds2d = b.ds[l].view(P, co // 2)
                elif self._side_grad(b, l) is None:
                    ds2d = dfm2d[:, off:off + co // 2]
                else:
                    ops.upsample_bwd(b.dfm, off, h, w, co // 2, out=b.ds[l])
                    ds2d = b.ds[l].view(P, co // 2)
                T.end(tok, 4.0 * B * ((h * w * (co // 2) + H * W + Kmax * (co // 2)) if self.fuse_pool_bwd else H * W * (co // 2)))
                ds2ds[l] = ds2d
                grp = b.groups[b.group_of[l]] if b.group_of[l] is not None else None
                if l >= lowest:              # G_l is only needed by backbone layers that train
                    tok = T.begin('side_bwd')
                    ops.gemm_nt(ds2d, pk.sideT[l], None, out=b.G[l].view(P, co))
                    T.end(tok, 2.0 * P * co * (co // 2))
                    if self.two_streams:
                        g_ready[l] = self.SLOT_G + l
                        ops.sync_record(g_ready[l])
            # The side convs' own weight gradients are parameter gradients nobody waits for before the optimiser, while the
            # dgrad chain waits for every G_l: all the G_l first (13 GEMMs), the weight gradients behind them.
            late_at = self._DEEP_SIDE_WGRAD_AT if self.two_streams else None
            if late_at is not None and not (lowest < late_at <= 12):      # the main loop below never reaches such a layer
                late_at = None
            late_side = [l for l in range(12, -1, -1) if late_at is not None and b.group_of[l] is not None]
            for l in range(12, -1, -1):
                if l not in late_side:
                    side_wgrad(l)
        # ---- main path, conv5_3 down to conv1_1.  The dgrad chain stays on the caller's stream; each layer's wgrad
        # (which only produces parameter gradients) goes to a third stream so that it fills the tails of the dgrad
        # kernels instead of sitting on the critical path.
        main = torch.cuda.current_stream()
        wg = self._wg() if self.two_streams else None
        for l in range(12, lowest - 1, -1):
            ci, co = CONV_CH[l]
            h, w = b.dims[l]
            idx = CONV_IDX[l]
            if g_ready[l] is not None:
                ops.sync_wait(g_ready[l], main.cuda_stream)
            if late_side and l == late_at:
                # the deep layers' side-conv weight gradients (TN products with K = pixels: MFMA-bound) only now, beside the
                # memory-bound transforms and 64-channel products of the last layers instead of beside conv5 / conv4
                with self._OnSide(self):
                    for ls_ in late_side:
                        side_wgrad(ls_)
                late_side = []
            x_in, relu_x = b.x_in[l], b.x_relu[l]      # what the forward of this layer read
            # one pass over G_l for both consumers (F(4x4) input gradient and weight gradient)
            dual = (self.dual_transform and b.wino_fwd[l] == 4 and l > lowest and trainable[l] and self.wgrad_winograd
                    and b.V[l] is not None and 'wgrad' not in self._diag_skip and ops.winograd_bias_rows(B, h, w, co) > 0
                    and not (POOL_AFTER[l - 1] and not self.fuse_unpool))
            v_dy = None
            if dual:
                Tl = ops.winograd_tiles(B, h, w, 4)
                if b.dV is None:
                    b.dV = torch.empty(max(36 * ops.winograd_tiles(B, *b.dims[i], 4) * CONV_CH[i][1] for i in range(1, 13)),
                                       dtype=torch.float32, device=self.device)
                if b.dM[l] is None:
                    b.dM[l] = torch.empty(36, Tl, co, dtype=torch.float32, device=self.device)
                    b.bpart[l] = torch.empty(ops.winograd_bias_rows(B, h, w, co), co, dtype=torch.float32, device=self.device)
                v_dy = b.dV[:36 * Tl * co].view(36, Tl, co)
                tok = T.begin('winograd_transform')
                if 'dual' not in self._diag_skip:
                    ops.winograd_dual_transform(b.G[l], v_dy, b.dM[l], b.bpart[l])
                T.end(tok, 4.0 * (B * h * w + 2 * 36 * Tl) * co)
            def wgrad(ws_tag):
                tok = T.begin('conv3x3_wgrad')
                dw, db = g[f'backbone.{idx}.weight'], g[f'backbone.{idx}.bias']
                mw = b.wino_fwd[l]                             # this forward went through the Winograd domain: its V was kept
                v_pre = b.V[l] if mw else None
                if dual:
                    ops.conv3x3_wgrad_winograd_pre(v_pre, b.dM[l], b.bpart[l], B, h, w, dw, db, ws_tag=ws_tag)
                    T.end(tok, 2.0 * 36 * ops.winograd_tiles(B, h, w, 4) * ci * co)
                    ready([f'backbone.{idx}.weight', f'backbone.{idx}.bias'])
                    return
                # the forward's kept V decides; without one (direct forward) only the layers where a transform pass of
                # its own still pays
                if self.wgrad_winograd and (v_pre is not None or (ci >= self.WINOGRAD_MIN_CI and co >= self.WINOGRAD_MIN_CO)):
                    mw = mw or self.WINOGRAD_TILE
                    ops.conv3x3_wgrad_winograd(x_in, b.G[l], relu_in=relu_x, dw=dw, db=db, ws_tag=ws_tag, v_pre=v_pre, m=mw)
                    # the FLOPs the MFMA pipe executes: (m+2)^2 positions x (m x m tiles) instead of 9 taps x pixels
                    T.end(tok, 2.0 * ops.winograd_positions(mw) * ops.winograd_tiles(B, h, w, mw) * ci * co)
                else:
                    ops.conv3x3_wgrad(x_in, b.G[l], ci, relu_in=relu_x, dw=dw, db=db, ws_tag=ws_tag)
                    T.end(tok, 2.0 * B * h * w * ci * co * 9)
                ready([f'backbone.{idx}.weight', f'backbone.{idx}.bias'])
            # With its operands ready (dual transform) a weight gradient can start any time.  Queued behind the layer's input
            # gradient instead of in front of it, its TN products run beside the NEXT layer's (memory-bound) transform rather
            # than beside this layer's products: 9.35 -> 9.20 ms.
            late_wgrad = wg is not None and dual and l > lowest + self._WGRAD_EARLY_LAYERS
            if l == lowest and wg is not None and self.on_tail is not None and trainable[l]:
                # every gradient but this layer's is queued (conv and fc weight gradients on wg, the side convs' on the side
                # stream): the step runner puts the bulk of the optimiser step on wg here, beside the last input gradient
                self.on_tail(wg, [f'backbone.{idx}.weight', f'backbone.{idx}.bias'])
            if not trainable[l] or 'wgrad' in self._diag_skip:
                pass
            elif late_wgrad:
                pass
            elif wg is not None:
                self._edge(main, wg)                       # G_l is final here
                with self._On(wg):
                    wgrad('wgrad')
            else:
                wgrad('default')
            if l > lowest:
                if g_ready[l - 1] is not None:
                    ops.sync_wait(g_ready[l - 1], main.cuda_stream)
                unpooled = False
                mbits = b.mbits[l - 1] if (self.compact_masks and b.mbits_ok[l - 1]) else None
                pcode = b.pcode[l - 1] if (self.compact_masks and b.pcode_ok[l - 1]) else None
                if gat[l - 1]:
                    pooled = POOL_AFTER[l - 1]
                    ops.conv3x3_dgrad_winograd_gather(b.G[l], pk.ud[l], b.dybar[l - 1], meta.new_row, None, out=b.G[l - 1],
                                                      mask_src=None if pooled else b.y[l - 1],
                                                      unpool_src=b.y[l - 1] if pooled else None, ws_tag='wino_main', timer=T,
                                                      mask_bits=None if pooled else mbits, unpool_code=pcode if pooled else None,
                                                      v_pre=v_dy)
                    unpooled = True
                elif b.wino_fwd[l]:
                    if POOL_AFTER[l - 1] and self.fuse_unpool and b.wino_fwd[l] == 4:
                        ops.conv3x3_dgrad_winograd_unpool(b.G[l], pk.ud[l], b.y[l - 1], b.G[l - 1], ws_tag='wino_main', timer=T,
                                                          unpool_code=pcode, v_pre=v_dy)
                        unpooled = True
                    elif POOL_AFTER[l - 1]:
                        ops.conv3x3_dgrad_winograd(b.G[l], pk.ud[l], out=b.dxp[l - 1], ws_tag='wino_main', timer=T,
                                                   m=b.wino_fwd[l], v_pre=v_dy)
                    else:
                        ops.conv3x3_dgrad_winograd(b.G[l], pk.ud[l], mask_src=None if mbits is not None else b.y[l - 1], out=b.G[l - 1],
                                                   accumulate=True, ws_tag='wino_main', timer=T, m=b.wino_fwd[l], mask_bits=mbits,
                                                   v_pre=v_dy)
                else:
                    tok = T.begin('conv3x3_dgrad')
                    if POOL_AFTER[l - 1]:
                        ops.conv3x3_dgrad(b.G[l], pk.wd[l], ci, out=b.dxp[l - 1])
                    else:
                        ops.conv3x3_dgrad(b.G[l], pk.wd[l], ci, mask_src=b.y[l - 1], out=b.G[l - 1], accumulate=True)
                    T.end(tok, 2.0 * B * h * w * ci * co * 9)
                if POOL_AFTER[l - 1] and not unpooled:
                    ops.maxpool2_bwd(b.y[l - 1], b.dxp[l - 1], b.G[l - 1], accumulate=True)
                if late_wgrad and trainable[l] and 'wgrad' not in self._diag_skip:
                    self._edge(main, wg)
                    with self._On(wg):
                        wgrad('wgrad')
        if wg is not None:
            self._edge(wg, main)
        self._join_side()
        self.ctx = None
