"""Headline benchmark: training images/sec of the WESUP step on GlaS-shaped synthetic patches.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step is one full training iteration (BASELINE.json metric): superpixel
preprocessing from the label map -> forward -> loss (label propagation + CE) -> backward ->
gradient all-reduce (N > 1) -> SGD, on a batch of 4 images per GPU, 480x480, 576 superpixels each,
fp32, with inputs already resident in HBM.  SLIC is excluded (label maps are inputs).
Rank 0 prints ONE JSON line.

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment launches itself: the parent (which never touches the GPU)
starts N fresh child processes of this script, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set, waits for them and exits non-zero if any of them does.  Under ``torch.distributed.run`` the
environment is already there and every process is a rank.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step uses 3 streams; a torch.distributed
# process group brings its own, and with 4 queues two of the step's streams then share one (false serialisation:
# +1.2 ms per step with nothing but the process group alive).  6 keeps every stream on a queue of its own; set before
# the HIP runtime starts.  
os.environ.setdefault('GPU_MAX_HW_QUEUES', '6')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md, Peak FP32 (matrix), dense
PEAK_HBM_GBS = 8000.0             # HBM3E spec; 6290 GB/s measured streaming copy


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='images per GPU (BASELINE configs[1]: 4)')
    ap.add_argument('--size', type=int, default=480)
    ap.add_argument('--grid', type=int, default=24, help='superpixel grid side: g*g superpixels per image')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--direct-conv', action='store_true', help='A/B: forward and input gradient of every conv layer with '
                                                               'the direct implicit-GEMM kernel (no Winograd-domain conv)')
    ap.add_argument('--direct-wgrad', action='store_true', help='A/B: every conv weight gradient with the direct '
                                                                'implicit-GEMM kernel (no Winograd-domain wgrad)')
    ap.add_argument('--engine-set', default='', help='A/B: comma list of name=value engine switches (bool / int attributes of WesupEngine), e.g. plain=1 (the reference\'s order of operations, one launch per pass)')
    ap.add_argument('--trainer-set', default='', help='A/B: comma list of name=value trainer kwargs (fuse_head, split_sgd, gc_freeze, trust_first_recording_after, plan_audit_every; "none" = None)')
    ap.add_argument('--diag-skip', default='', help="TIMING-ONLY diagnostic (results are wrong): comma list of launch classes left out "
                                                    "of the step after the warm-up (learning rate 0 from there on) -- 'wgrad' (conv weight gradients), 'side_wgrad', "
                                                    "'side_fwd_shallow' (pooling + side conv of conv1_1 .. conv3_3), 'side_fwd_deep', 'tout' (output transforms of the K = 512 layers), 'tin' (forward input transforms), 'dual' (the backward's dual transforms) -- to see what they cost the step")
    ap.add_argument('--event-every', type=int, default=10, help='steps of the timed region that carry HIP events: every n-th')
    ap.add_argument('--timed-classes', default='conv3x3_fwd,conv3x3_dgrad,winograd_gemm',
                    help="kernel classes that get HIP events inside the timed region ('all', 'none' or a comma list); an "
                         "event record fences its queue, so only the dominant kernel is timed there by default and the "
                         "other classes are timed in extra untimed steps")
    ap.add_argument('--unfused-pool-bwd', action='store_true')
    ap.add_argument('--force-ddp', action='store_true', help='run the RCCL gradient all-reduce path even with one rank')
    ap.add_argument('--bucket-mb', type=int, default=16, help='gradient all-reduce bucket size')
    ap.add_argument('--rccl-channels', type=int, default=0, help='A/B on a multi-GPU box: NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = this '
                    '(RCCL kernels occupy one workgroup per channel: the CUs they take from the step); 0 = RCCL\'s own choice')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='multi-rank rehearsal on a one-GPU box: every rank uses cuda:0 and gloo carries the exchange '
                         '(RCCL refuses two ranks on one device); exercises the launch contract, not performance')
    ap.add_argument('--ddp-probe', default='', help="diagnostics with one rank: 'pg' = process group only, 'reducer' = reducer without the collective")
    ap.add_argument('--end-to-end', action='store_true', help='second bench line (never the headline): decoded uint8 batch -> wesup_augment -> '
                    'GPU SLIC at sp_area 200, one batch ahead on a second stream -> training step; same JSON schema')
    ap.add_argument('--multiscale', type=int, default=0, help='second bench line: the reference\'s own operating point (models/wesup.py:178, '
                    'utils/data.py:98-101): batch 1, a new (H, W) every step drawn from 0.3-0.4 x 775x522, this many distinct shapes in rotation')
    ap.add_argument('--no-step-plan', action='store_true', help='A/B: every iteration walks the launch list in Python (no replay of a recorded step plan, wesup_amd/runner.py)')
    ap.add_argument('--general-path', action='store_true', help='A/B: the trainer\'s general path (preprocess / forward / compute_loss / autograd backward) instead of the step runner')
    ap.add_argument('--stub-trainer', action='store_true',
                    help='launch-contract rehearsal WITHOUT a GPU: every rank runs a trivial CPU step under gloo; only '
                         'the launcher, the rendezvous, the barrier/max-over-ranks timing and the one JSON line are real '
                         '(tests/test_bench_launch_cpu.py); the value it prints measures nothing')
    ap.add_argument('--stub-fail-rank', type=int, default=-1, help='with --stub-trainer: this rank exits with an error')
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------- self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """Parent of ``python bench.py --gpus N`` (N > 1, no torch.distributed.run around it): N fresh child processes,
    one rank each.  Nothing here initialises the GPU (no HIP call, no exec of a process that did)."""
    port = _free_port()
    children = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(children)
        while pending:
            for ch in list(pending):
                r = ch.poll()
                if r is None:
                    continue
                pending.remove(ch)
                if r != 0 and rc == 0:
                    rc = r if r > 0 else 1
                    for other in pending:          # a rank died: the others would wait in a collective forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for ch in children:
            if ch.poll() is None:
                ch.kill()
    return rc


# --------------------------------------------------------------------------------------------- helpers
def workload_label(B, H, W, g):
    """Names the BASELINE.json config the shape corresponds to, or says that it is none of them."""
    n = g * g
    known = {(480, 480, 4, 576): 'BASELINE configs[1] (per-GPU shard of configs[2]): GlaS-shaped',
             (800, 800, 4, 1521): 'per-GPU shard of BASELINE configs[3]: CRAG-shaped',
             (1024, 1024, 8, 3025): 'per-GPU shard of BASELINE configs[4]: H&E-shaped'}
    head = known.get((H, W, B, n), 'custom shape (not a BASELINE config):')
    return (f'{head} synthetic {H}x{W} patches, VGG16 side-output extractor, batch={B}/GPU, {n} superpixels/img '
            '(jittered Voronoi), 20% point-labelled, full train step (preprocess+fwd+loss+bwd+allreduce+SGD), '
            'SLIC excluded')


def roofline_inputs(B, H, W, g):
    """Counter-derived inputs of the roofline objects, read from the newest profiles/rNN_roofline_inputs.json
    (written by tools/roofline_inputs.py from the rocprofv3 --pmc passes of tools/collect_profiles.sh).  Only used
    when the file was collected at this very shape; no number is typed into this file."""
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*roofline_inputs.json')), reverse=True):
        with open(path) as f:
            d = json.load(f)
        sh = d.get('shape', {})
        if (sh.get('batch'), sh.get('H'), sh.get('W'), sh.get('superpixels')) == (B, H, W, g * g):
            d['file'] = os.path.relpath(path, ROOT)
            return d
    return None


def host_cpu():
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:
        cpu = 'unknown CPU'
    return f'{cpu}, {os.cpu_count()} logical cores'


def stub_worker(args, rank, world):
    """--stub-trainer: the launch contract on CPU ranks (gloo).  See the flag's help."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo')
    if rank == args.stub_fail_rank:
        raise SystemExit(3)
    x = torch.ones(1024)

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(0.002)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))         # ranks differ: the reported time must be the slowest rank's
        if world > 1:
            dist.all_reduce(x)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = tmin = elapsed
    if world > 1:
        t = torch.tensor([elapsed, -elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax, tmin = float(t[0]), -float(t[1])
    if rank == 0:
        print(json.dumps({'metric': 'STUB (launch-contract rehearsal, measures nothing)', 'value': world * args.batch * args.steps / tmax,
                          'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': tmax / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'none (stub)',
                          'config': {'workload': 'stub', 'parallelism': f'dp{world}'},
                          'rank_time': {'min_s': tmin, 'max_s': tmax},
                          'collective': {'backend': 'gloo', 'ranks': world, 'allreduce_sum': float(x[0])}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------- one rank
def rccl_version():
    try:
        import torch
        v = torch.cuda.nccl.version()
        return '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:       # a CPU rehearsal (gloo): no RCCL in the process
        return f'unavailable ({type(e).__name__})'


def worker(args):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    if args.stub_trainer:
        return stub_worker(args, rank, world)
    if args.multiscale or args.end_to_end:            # second lines: throughput only (the roofline objects belong to the headline)
        args.no_kernel_timing = args.no_cpu_baseline = True
    # stdout carries ONE line, the JSON of rank 0: whatever a library prints there while the step runs (RCCL announces its
    # version on stdout when the first communicator is created) goes to stderr instead -- at the descriptor level, native
    # writers included -- and the line is written after the descriptor is back
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np  # noqa: F401
    import torch
    use_dist = world > 1 or args.force_ddp or bool(args.ddp_probe)
    backend = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.rccl_channels > 0:                    # read by RCCL when the communicator is created
            os.environ['NCCL_MIN_NCHANNELS'] = os.environ['NCCL_MAX_NCHANNELS'] = str(args.rccl_channels)
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        backend = dist.get_backend()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from oracle import wesup_oracle as orc          # only for the seeded weights + the cpu_baseline leg
    from wesup_amd import synth
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice

    B, H, W, g = args.batch, args.size, args.size, args.grid
    weights = orc.make_weights(0, feat_scale=0.05)
    tkw = {}
    if args.multiscale:        # a new shape every other step: startup objects out of the collector, clean first recordings sealed (and audited)
        tkw.update(gc_freeze=True, trust_first_recording_after=1)
    tkw.update({k: (None if v == 'none' else int(v)) for k, v in (kv.split('=') for kv in filter(None, args.trainer_set.split(',')))})
    trainer = initialize_trainer('wesup', device=str(dev), max_superpixels=g * g, force_allreduce=args.force_ddp,
                                 step_plan=not args.no_step_plan, native_step=not args.general_path, **tkw)
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    trainer.optimizer, trainer.scheduler = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.model.train()
    trainer.tracker.train()
    eng = trainer.model.engine
    eng.fuse_pool_bwd = not args.unfused_pool_bwd
    eng.conv_winograd = not args.direct_conv
    eng.wgrad_winograd = not args.direct_wgrad
    for kv in filter(None, args.engine_set.split(',')):
        k, v = kv.split('=')
        assert isinstance(getattr(eng, k), (bool, int)), k
        setattr(eng, k, type(getattr(eng, k))(int(v)))
    if use_dist and args.ddp_probe != 'pg':
        trainer.enable_data_parallel(bucket_bytes=args.bucket_mb << 20)

    # two different synthetic batches per rank, resident in HBM before the timed region
    pool = []
    e2e = None
    if args.multiscale:
        # the reference trains at batch 1 on GlaS images (775 x 522) rescaled by U(0.3, 0.4) per item: a new shape almost every step
        rs = np.random.RandomState(7 + rank)
        B = 1
        for i in range(args.multiscale):
            f = rs.uniform(0.3, 0.4)
            h, w = int(522 * f), int(775 * f)
            gi = max(2, int(round((h * w / 200.0) ** 0.5)))           # sp_area 200 (models/wesup.py:158)
            imgs, labs, pts, pix = synth.make_batch(1000 * rank + i + 1, 1, h, w, gi)
            pool.append((torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev),
                         torch.from_numpy(labs).to(dev)))
        trainer.kwargs['max_superpixels'] = None                      # rows = the label map's own count (known on the host)
        from wesup_amd.utils.data import LabelMaps as _LM
        pool = [(a, b_, c, _LM(d, [int(d.max()) + 1])) for a, b_, c, d in pool]
    elif args.end_to_end:
        from wesup_amd import ops as _ops_e
        from wesup_amd.utils import data as _D
        rs = np.random.RandomState(11 + rank)
        trainer.kwargs['max_superpixels'] = None
        seg_fn = trainer.prefetch_segment_fn()
        side = torch.cuda.Stream(device=dev)
        raw = []
        for i in range(2):
            u8 = np.ascontiguousarray((np.stack([synth.synth_image(1000 * rank + 10 * i + b_, H, W) for b_ in range(B)])
                                       .transpose(0, 2, 3, 1) * 255).astype(np.uint8))
            msk = (rs.random_sample((B, H, W)) > 0.5).astype(np.uint8)
            par = np.stack([_D.sample_params(rs, H, W, True)[0] for _ in range(B)])
            pts = torch.zeros(B, 2, H, W, dtype=torch.uint8, device=dev)
            idx = rs.randint(0, min(H, W), (B, 120, 2))
            for b_ in range(B):
                pts[b_, rs.randint(0, 2, 120), idx[b_, :, 0], idx[b_, :, 1]] = 1
            raw.append((torch.from_numpy(u8).to(dev), torch.from_numpy(msk).to(dev), torch.from_numpy(par).to(dev), pts))
        e2e = {'i': 0}

        def stage(i):
            d_img, d_mask, params, pts = raw[i % 2]
            with torch.cuda.stream(side):
                img, pm = _ops_e.augment(d_img, d_mask, params)
                seg, n_dev = seg_fn(img)
                counts = torch.empty(n_dev.shape, dtype=n_dev.dtype).pin_memory()
                counts.copy_(n_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            return img, pm, pts, seg, counts, ev
        e2e['nxt'] = stage(0)
    else:
        for i in range(2):
            imgs, labs, pts, pix = synth.make_batch(1000 * rank + i + 1, B, H, W, g)
            pool.append((torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev),
                         torch.from_numpy(labs).to(dev)))


    def step(i):
        if e2e is not None:
            # augmentation + SLIC of the NEXT batch run on a second stream beside this step; its superpixel counts reached
            # pinned host memory meanwhile (utils/data.py DevicePrefetcher does the same from a DataLoader)
            img, pm, pts, seg, counts, ev = e2e['nxt']
            e2e['nxt'] = stage(i + 1)
            torch.cuda.current_stream().wait_event(ev)
            ev.synchronize()
            for t_ in (img, pm, seg):
                t_.record_stream(torch.cuda.current_stream())
            return trainer.train_one_iteration('train', img, pm, pts, _D.LabelMaps(seg, [int(v) for v in counts]))
        return trainer.train_one_iteration('train', *pool[i % len(pool)])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # What THIS box delivers, measured before the run (the pool's boxes differ by +-2.5 % for the same code): a 256 MiB
    # device-to-device copy and one large plain NT GEMM of the library (4096^3), each alone on the GPU.  Context for comparing
    # lines across runs, not part of any roofline figure.
    box = None
    if rank == 0 and not args.no_kernel_timing:
        from wesup_amd import ops as _ops_b
        src_ = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        dst_ = torch.empty_like(src_)
        A_ = torch.randn(4096, 4096, device=dev)
        B_ = torch.randn(4096, 4096, device=dev)
        C_ = torch.empty(4096, 4096, device=dev)

        def _probe(fn, reps):
            fn(); torch.cuda.synchronize()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0_.record()
            for _ in range(reps):
                fn()
            e1_.record(); torch.cuda.synchronize()
            return e0_.elapsed_time(e1_) / reps * 1e-3
        t_copy = _probe(lambda: dst_.copy_(src_), 20)
        t_gemm = _probe(lambda: _ops_b.gemm_nt(A_, B_, None, out=C_), 5)
        box = {'copy_256MiB_gbs': round(2 * (256 << 20) / t_copy / 1e9, 1), 'gemm_nt_4096_cubed_tflops': round(2 * 4096.0 ** 3 / t_gemm / 1e12, 1),
               'what': 'this box alone, before the run: device-to-device copy (read + write bytes / time) and wesup_gemm_nt 4096 x 4096 x 4096'}
        del src_, dst_, A_, B_, C_
        torch.cuda.empty_cache()

    timer = trainer.model.engine.timer
    from wesup_amd import ops as _ops_t
    _ops_t.set_timer(timer)                           # sp_preprocess / propagate / paint / sgd launch outside the engine
    timer.reset()
    timing_on = not args.no_kernel_timing and args.timed_classes != 'none'
    timer.only = None if args.timed_classes == 'all' else set(args.timed_classes.split(','))
    n_sampled = 0
    reducer = getattr(trainer, 'reducer', None)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # one record per step: its end
    if timing_on:
        # The event-carrying steps of the timed region take their events from the timer's free list: made (and recorded once) here.
        # Creating them inside the region made the runtime grow its signal pool there -- one step of 79 ms in a 60-step batch-1 run
        # (round 5, `WESUP_BENCH_STEPS=1`), i.e. a third of the line's img/s.
        pre = [torch.cuda.Event(enable_timing=True) for _ in range((args.steps // max(1, args.event_every) + 1) * 1000)]
        for e_ in pre:
            e_.record()
        torch.cuda.synchronize()
        timer._free.extend(pre)
    # The warm-up directly in front of the timed region: with the event pool made between the two (15 ms of host work and a
    # device sync) the GPU idled long enough to leave its clocks: the region's first two steps took 8.9 / 8.4 ms against 8.0.
    for i in range(args.warmup):
        step(i)
    if args.diag_skip:            # TIMING-ONLY: classes of launches left out from here on (buffers keep the warm-up's values)
        trainer.model.engine._diag_skip = set(filter(None, args.diag_skip.split(',')))
        _ops_t.DIAG = set(trainer.model.engine._diag_skip) & {'tin', 'tout'}
        for gr in trainer.optimizer.param_groups:
            gr['lr'] = 0.0
    ev_off = args.event_every // 2 if args.steps > args.event_every // 2 else 0      # a short run (--steps 5) still has one such step
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # an event record fences its queue (events on the 25 conv launches of every step cost 3 % of the step, on every
        # kernel class 16 %): inside the timed region only every --event-every'th step carries events
        timer.enabled = timing_on and i % args.event_every == ev_off      # (not the region's first step where there is another)
        n_sampled += int(timer.enabled)
        if reducer is not None:                       # bucket launch offsets + exposed all-reduce tail of the same steps
            reducer.profile = (i % args.event_every == ev_off)
        step(i)
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    step_ms_raw = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get('WESUP_BENCH_STEPS'):
        print('per-step ms:', ' '.join(f'{v:.2f}' for v in step_ms_raw), file=sys.stderr)
    step_ms = sorted(step_ms_raw)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    ddp_stats = None
    if reducer is not None:
        reducer.profile = False
        ddp_stats = reducer.collect_stats()
    rank_time = None
    if use_dist:
        t = torch.tensor([elapsed, -elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rank_time = {'min_s': round(-float(t[1].item()), 4), 'max_s': round(float(t[0].item()), 4)}
        elapsed = float(t[0].item())                      # the slowest rank's wall time of the timed region
    loss_last = trainer.tracker.history['loss'][-1]
    runner = trainer.step_runner()
    plan_stats = dict(runner.stats) if runner is not None else None      # iterations of warm-up + timed region by how they ran
    if runner is not None and os.environ.get('WESUP_PLAN_DEBUG'):
        import collections
        print('[step plan] states', len(runner.states), collections.Counter((st.plan is not None, st.cand is not None, st.count, st.tries) for st in runner.states.values()), file=sys.stderr)

    # ---- untimed extras for the roofline report (every rank runs them so that collectives stay matched)
    iso, pool_ms = None, None
    if not args.no_kernel_timing:
        eng = trainer.model.engine
        timed = dict(timer.collect())                 # keep the timed-region numbers
        timer.reset()
        timer.only = None
        timer.enabled = True
        n_extra = 4
        for i in range(n_extra):                      # same schedule, every kernel class with events
            step(i)
        torch.cuda.synchronize()
        allk = dict(timer.collect())
        timer.reset()
        eng.two_streams = False                       # kernels alone on the GPU: isolated per-launch durations
        from wesup_amd import ops as _ops
        for i in range(3):                            # (first step: workspaces are allocated)
            if i == 1:
                torch.cuda.synchronize()
                timer.collect(); timer.reset()
            step(i)
        torch.cuda.synchronize()
        iso = dict(timer.collect())
        timer.enabled = False
        eng.two_streams = True
        timer.reset()
        timer.totals = timed
        if rank == 0:
            from wesup_amd import ops
            fm = eng.feature_maps()                   # materialise the (B,H,W,2112) map of the last step
            meta = trainer.model._last_meta
            outp = torch.empty(B, meta.Kmax, fm.shape[-1], device=dev)
            ops.sp_pool_fwd(fm, meta, out=outp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.sp_pool_fwd(fm, meta, out=outp)
            e1.record()
            torch.cuda.synchronize()
            pool_ms = e0.elapsed_time(e1) / 5

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {
            'metric': 'training images/sec, GlaS 480x480 VGG16 ~600 SP/img', 'value': round(value, 3), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'ms_per_step_median': round(median_ms, 3),      # of the per-step GPU times (an event at the end of every step)
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload_label(B, H, W, g),
                       'global_batch': world * B, 'image': [H, W], 'superpixels': g * g,
                       'parallelism': f'dp{world}', 'last_loss': loss_last},
            # how the warm-up + timed iterations were issued (wesup_amd/runner.py): walked in Python ('eager'; the first two of a
            # shape and every event-carrying one), walked and recorded, or replayed from the recorded step plan
            'step_plan': plan_stats,
            'box': box,
            'memory': {'peak_allocated_gib': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                       'reserved_gib': round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2),
                       'cached_buffer_sets': len(trainer.model.engine._bufs)},
        }
        if args.end_to_end:
            out['config']['workload'] = ('END-TO-END (second line, not the headline): decoded uint8 batch resident in HBM -> wesup_augment '
                                         '(flips, shift-scale-rotate, HSV, brightness/contrast, ToTensor, one-hot) -> GPU SLIC at sp_area 200 '
                                         '(wesup_slic, one batch ahead on a second stream, counts through pinned memory) -> ' + out['config']['workload'])
            out['config']['superpixels'] = 'SLIC, sp_area 200'
        if args.multiscale:
            shapes = sorted({tuple(p_[0].shape[2:]) for p_ in pool})
            out['config']['workload'] = (f'MULTI-SCALE (second line, not the headline): the reference\'s own operating point -- batch 1, GlaS 775x522 '
                                         f'rescaled by U(0.3, 0.4) per item, {len(shapes)} distinct shapes in rotation ({shapes[0]} ... {shapes[-1]}), '
                                         'sp_area 200 Voronoi label maps, 20% point-labelled, full train step, SLIC excluded')
            out['config']['image'], out['config']['superpixels'] = 'varies', 'varies'
        rin = roofline_inputs(B, H, W, g)
        if not args.no_kernel_timing:
            tot = timer.collect()
            n_ev = n_sampled
            if not tot:                               # --timed-classes none: the roofline comes from the extra steps
                tot, n_ev = allk, n_extra
            kern = {}
            for tag, (ms, n, work) in sorted(allk.items()):
                kern[tag] = {'ms_per_step': round(ms / n_extra, 4), 'launches_per_step': n / n_extra,
                             'avg_us': round(ms / n * 1e3, 2)}
                if tag.startswith('conv3x3') or tag in ('winograd_gemm', 'side_fwd', 'side_bwd', 'mlp_fwd', 'mlp_bwd', 'mlp_wgrad', 'sp_pool_mat_fwd', 'upsample_mat_bwd'):
                    kern[tag]['tflops'] = round(work / (ms * 1e-3) / 1e12, 2)
                elif work > 0:       # memory-bound classes: algorithmic bytes / event time, against the HBM peak
                    kern[tag]['gbs'] = round(work / (ms * 1e-3) / 1e9, 1)
                    kern[tag]['frac_of_hbm_peak'] = round(work / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            if iso is not None:      # the same classes alone on the GPU (single-stream steps)
                for tag, (ms, n, work) in iso.items():
                    if tag in kern and ms > 0:
                        kern[tag]['alone_ms_per_step'] = round(ms / 2, 4)
                        if 'gbs' in kern[tag] and work > 0:
                            kern[tag]['alone_gbs'] = round(work / (ms * 1e-3) / 1e9, 1)
            kern['_event_pairs_replaced'] = timer.replaced      # KernelTimer.collect: impossible durations (timestamp glitches) ...
            kern['_event_pairs_replaced_list'] = [list(x) for x in timer.replaced_pairs]      # ... (class, measured ms, substituted median ms)
            kern['_note'] = ('propagate = wesup_propagate (label propagation); sp_preprocess = wesup_sp_preprocess + '
                             'wesup_sp_segments (histograms, reference ordering, counting sort, segment table); paint = '
                             'wesup_paint_fwd; sgd = wesup_sgd_step (20 B per parameter).  gbs = algorithmic bytes / event time')
            # Dominant kernel: gemm_nt_kernel, the GEMM of every conv forward and input gradient -- the implicit-GEMM form
            # (MODE 1/2) for the 3/64-channel layers, the 16-position batched form (MODE 3) over Winograd-domain operands
            # for the layers with >= 128 input channels, whose two memory-bound transform passes per op belong to the same
            # work.  `achieved` follows SURVEY.md 8(d): ALGORITHMIC FLOPs of the conv forward + input gradient (direct form,
            # 2*H*W*Cin*Cout*9 per layer and pass) over the event time of every launch that carries them out (GEMMs and
            # transforms).  `mfma_executed` is the matrix-core view of the GEMM launches alone: the FLOPs the MFMA pipe
            # executes (the Winograd form needs 4/9 of the direct form's) over their event time.
            gemm_tags = ('conv3x3_fwd', 'conv3x3_dgrad', 'winograd_gemm')
            if any(t in allk and t not in tot for t in gemm_tags):          # a GEMM class was left out of --timed-classes
                tot, n_ev = allk, n_extra
            ms_gemm = sum(tot[t][0] for t in gemm_tags if t in tot)
            fl_exec = sum(tot[t][2] for t in gemm_tags if t in tot)
            nl = sum(tot[t][1] for t in gemm_tags if t in tot)
            # the transform launches (effective_direct_form only): from the timed region when --timed-classes names them, else
            # from the extra steps (events around them as well cost the event-carrying steps another 0.5 ms each)
            if 'winograd_transform' in tot:
                ms_tr = tot['winograd_transform'][0]
            else:
                ms_tr = allk.get('winograd_transform', (0.0, 0, 0.0))[0] / n_extra * n_ev
            ms_all = ms_gemm + ms_tr
            fl_alg, hh, ww = 0.0, H, W
            for l, (ci, co) in enumerate(((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
                                          (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512))):
                fl_alg += 2.0 * B * hh * ww * ci * co * 9 * (2 if l > 0 else 1)          # conv1_1 has no input gradient
                if l in (1, 3, 6, 9):
                    hh, ww = hh // 2, ww // 2
            eff = fl_alg * n_ev / (ms_all * 1e-3) / 1e12
            ach_exec = fl_exec / (ms_gemm * 1e-3) / 1e12
            conv_in = (rin or {}).get('conv3x3_fwd_dgrad', {})
            # `achieved` / `frac`: what is compared with the hardware peak -- the FLOPs the MFMA pipe EXECUTES in the GEMM
            # launches over their event time.  The direct-form (SURVEY 8(d) algorithmic) FLOPs over GEMM + transform time are an
            # effective rate of the whole op (the Winograd forms execute 4/9 resp. 1/4 of them; it can exceed the peak) and
            # carry no fraction of the peak.
            out['roofline'] = {'bound': 'mfma',
                               'kernel': 'the MFMA launches of conv3x3 fwd+dgrad: gemm_nt_kernel (implicit GEMM for the image layer; batched '
                                         'GEMMs over Winograd F(4x4,3x3)-domain operands for the products with K = 512; fp32 MFMA '
                                         '32x32x2) and wino4_gemm_out_kernel (products with K <= 256: the 36 products AND the output '
                                         'transform in one kernel; fp32 MFMA 16x16x4)',
                               'achieved': round(ach_exec, 2), 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(ach_exec / PEAK_MFMA_F32_TFLOPS, 4),
                               'traffic': conv_in.get('hbm_bytes_per_launch'),
                               'avg_launch_us': round(ms_gemm / nl * 1e3, 2), 'launches_per_step': nl / n_ev,
                               'flop_per_step': fl_exec / n_ev, 'event_timed_steps': n_ev,
                               'what': 'FLOPs the MFMA pipe executes in the conv forward + input-gradient MFMA launches (rocprofv3 --stats: '
                                       'gemm_nt_kernel<..,1|2|3,..> and wino4_gemm_out_kernel<..>, whose time includes the fused output '
                                       'transform) / their HIP-event time inside the timed region',
                               'ms_per_step': {'gemm': round(ms_gemm / n_ev, 3), 'transforms': round((ms_all - ms_gemm) / n_ev, 3)},
                               'effective_direct_form': {'tflops': round(eff, 2), 'flop_per_step': fl_alg,
                                                         'what': 'SURVEY 8(d) algorithmic (direct-form 2*H*W*Cin*Cout*9) FLOPs of '
                                                                 'the same ops / event time of their GEMM AND transform launches: an '
                                                                 'effective rate of the ops, not a utilisation of the MFMA pipe'}}
            if conv_in:
                out['roofline']['traffic_how'] = (f"{rin['file']}: rocprofv3 --pmc over this command, FETCH_SIZE x2 (gfx950 "
                                                  'correction) + WRITE_SIZE, separate passes, mean over the conv fwd+dgrad '
                                                  'launches of a step')
                out['roofline']['algorithmic_bytes_per_launch'] = conv_in.get('algorithmic_bytes_per_launch')
                out['roofline']['mfma_pipe_busy_frac'] = conv_in.get('mfma_busy_frac')
            if 'conv3x3_wgrad' in allk:
                ms, n, fl = allk['conv3x3_wgrad']
                a = fl / (ms * 1e-3) / 1e12
                out['roofline_wgrad'] = {'bound': 'mfma', 'kernel': 'conv3x3 wgrad ops: gemm_tn_kernel + reduce, implicit GEMM for conv1_1..conv2_1, '
                                                                    'Winograd-domain (outgrad transform + batched TN GEMMs + G^T.G reduce) above; '
                                                                    'executed FLOPs over the whole op',
                                         'achieved': round(a, 2), 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                                         'frac': round(a / PEAK_MFMA_F32_TFLOPS, 4),
                                         'how': f'{n_extra} extra untimed steps, same 3-stream schedule'}
                wg_in = (rin or {}).get('conv3x3_wgrad', {})
                if wg_in:
                    out['roofline_wgrad']['traffic'] = wg_in.get('hbm_bytes_per_launch')
                    out['roofline_wgrad']['mfma_pipe_busy_frac'] = wg_in.get('mfma_busy_frac')
            out['roofline']['note'] = (f'HIP events on the launch stream around every conv fwd/dgrad GEMM launch of {n_ev} of the '
                                       f'{args.steps} timed steps; the side-branch and wgrad streams run concurrently '
                                       '(roofline_isolated has the same kernels alone on the GPU)')
            out['kernels'] = kern
            # whole-step view of the matrix cores: every GEMM-shaped FLOP of the step (conv fwd/dgrad/wgrad, side convs,
            # MLP, matrix pooling of the deep layers) over the wall time of the step, all streams together
            gemm_tags = ('conv3x3_fwd', 'conv3x3_dgrad', 'winograd_gemm', 'conv3x3_wgrad', 'side_fwd', 'side_bwd', 'mlp_fwd', 'mlp_bwd', 'mlp_wgrad',
                         'sp_pool_mat_fwd', 'upsample_mat_bwd')
            fl_step = sum(allk[t][2] for t in gemm_tags if t in allk) / n_extra
            a = fl_step / (ms_per_step * 1e-3) / 1e12
            out['roofline_step'] = {'bound': 'mfma', 'what': 'all GEMM FLOPs one step EXECUTES (Winograd-domain passes: 1/4 resp. 4/9 of the direct form) / wall time of the step (3 streams)',
                                    'achieved': round(a, 2), 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(a / PEAK_MFMA_F32_TFLOPS, 4), 'flop_per_step': fl_step}
            out['kernels_how'] = (f'{n_extra} extra untimed steps with events on every kernel class (same 3-stream '
                                  'schedule); events on every class inside the timed region cost 16 % of the step')
            if iso is not None:
                def tf(tags):
                    ms = sum(iso[t][0] for t in tags if t in iso)
                    fl = sum(iso[t][2] for t in tags if t in iso)
                    return round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None
                a, wgr = tf(('conv3x3_fwd', 'conv3x3_dgrad', 'winograd_gemm')), tf(('conv3x3_wgrad',))       # executed FLOPs
                out['roofline_isolated'] = {
                    'how': '2 extra untimed steps with single-stream scheduling (the same kernels as the 3-stream step, each alone on the GPU), HIP events per launch',
                    'ms_per_step': {k: round(v[0] / 2, 3) for k, v in sorted(iso.items())},
                    'flops': 'executed (GEMM launches only; Winograd-domain passes run 1/4 resp. 4/9 of the direct form)',
                    'conv3x3_fwd_dgrad': {'bound': 'mfma', 'achieved': a, 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                                          'frac': round(a / PEAK_MFMA_F32_TFLOPS, 4)},
                    'conv3x3_wgrad': {'bound': 'mfma', 'achieved': wgr, 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                                      'frac': round(wgr / PEAK_MFMA_F32_TFLOPS, 4)},
                }
            if pool_ms is not None:
                by = 4.0 * B * (2112 * H * W + H * W + g * g * 2112)
                a = by / (pool_ms * 1e-3) / 1e9
                sm_in = (rin or {}).get('scatter_mean', {})
                # The scatter-mean the STEP runs leads: tile-form pooling of the native-resolution layers
                # (sp_pool_tile_kernel + combine), segment-form pooling of the coarse gather layers (sp_pool_up_fwd_kernel), the
                # interpolation-pooling matrix of the deep layers and its application (interp_matrix + sp_pool_mat_fwd), on ITS OWN
                # byte model; the materialised-map kernel of SURVEY 8(d) (never launched by the step) is the secondary figure.
                ftags = ('sp_pool_up_fwd', 'sp_pool_mat_fwd', 'interp_matrix')
                f_in = sum(allk[t][0] for t in ftags if t in allk) / n_extra
                f_alone = sum(iso[t][0] for t in ftags if t in iso) / 2 if iso is not None else None
                dims, hh, ww = [], H, W
                for l, co in enumerate((64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)):
                    dims.append((hh, ww, co // 2))
                    if l in (1, 3, 6, 9):
                        hh, ww = hh // 2, ww // 2
                own = 0.0
                eng_ = trainer.model.engine
                commuted = eng_.commute_side
                for (h_, w_, c_) in dims:
                    matrix = (h_, w_) != (H, W) and h_ * w_ <= 4096
                    # read once: the side output -- or, for the gather layers with the side conv commuted behind the pooling,
                    # the conv output itself (twice the channels; the side output is never formed)
                    own += 4.0 * h_ * w_ * (c_ if (matrix or not commuted) else 2 * c_)
                    if not matrix:              # gather layers: the sorted pixel list
                        own += 4.0 * H * W
                for (h_, w_) in sorted({(h_, w_) for (h_, w_, _) in dims if (h_, w_) != (H, W) and h_ * w_ <= 4096}):
                    own += 4 * 4.0 * g * g * h_ * w_                             # Wm and its transpose: written, then read
                own = B * (own + 4.0 * g * g * 2112)
                sm = {'bound': 'hbm',
                      'kernel': 'the scatter-mean of the training step: '
                                + 'sp_pool_up_fwd_kernel + sp_pool_combine_kernel (segment form: conv1_1 ... conv3_3, upsample fused)'
                                + ', sp_interp_matrix_kernel + gemm_tn (the six deep layers)',
                      'algorithmic_bytes': own,
                      'bytes_what': 'what the pooling reads at the layers\' own resolution (gather layers: the conv output, all C channels, the '
                                    'side conv being applied to the pooled rows; matrix layers: the side output) + slot bytes / pixel lists of '
                                    'the gather layers + the interpolation-pooling matrices (written and read) + N*2112*4',
                      'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'ms_per_step_in_step': round(f_in, 4),
                      'ms_per_step_alone': None if f_alone is None else round(f_alone, 4)}
                # primary: what the launches reach INSIDE the 3-stream step (the thing the bench times); `alone` = the same launches
                # by themselves on the GPU (single-stream extra steps)
                if f_in:
                    sm['achieved'] = round(own / (f_in * 1e-3) / 1e9, 1)
                    sm['frac'] = round(sm['achieved'] / PEAK_HBM_GBS, 4)
                    sm['achieved_how'] = 'HIP events around the launches inside the 3-stream step (they share the GPU with the other two streams); alone: `alone`'
                    sm['in_step'] = {'achieved': sm['achieved'], 'frac': sm['frac']}
                if f_alone:
                    sm['alone'] = {'achieved': round(own / (f_alone * 1e-3) / 1e9, 1), 'frac': round(own / (f_alone * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                by = 4.0 * B * (2112 * H * W + H * W + g * g * 2112)
                a = by / (pool_ms * 1e-3) / 1e9
                sm_in = (rin or {}).get('scatter_mean', {})
                sm['materialised'] = {
                    'kernel': 'sp_pool_fwd_kernel: the scatter-mean over a materialised (HW x 2112) feature map (SURVEY 8(d)\'s byte model: '
                              '2112*HW*4 + HW*4 + N*2112*4 per image); launched by bench.py after the timed region only',
                    'achieved': round(a, 1), 'frac': round(a / PEAK_HBM_GBS, 4), 'frac_of_measured_copy_6290': round(a / 6290.0, 4),
                    'traffic': sm_in.get('hbm_bytes_per_launch'),
                    'traffic_how': (f"{rin['file']}: FETCH_SIZE x2 + WRITE_SIZE of sp_pool_fwd_kernel in the PMC passes over "
                                    'this command') if sm_in else None,
                    'avg_launch_us': round(pool_ms * 1e3, 2), 'algorithmic_bytes': by,
                    'step_path_on_this_model': {k: round(by / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) for k, ms_ in
                                                (('frac_in_step', f_in), ('frac_alone', f_alone)) if ms_}}
                out['roofline_scatter_mean'] = sm
        if use_dist:
            out['rank_time'] = rank_time              # spread of the per-rank wall time of the timed region
            out['collective'] = {'backend': backend, 'ranks': dist.get_world_size(),
                                 'what': 'bucketed all-reduce(sum) of the flat fp32 gradient buffer, 1/world folded into SGD',
                                 'bucket_mb': args.bucket_mb, 'exposed_ms': None, 'buckets': None,
                                 # what the first multi-GPU run has to be read against
                                 'env': {k: os.environ.get(k) for k in ('GPU_MAX_HW_QUEUES', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS',
                                                                        'NCCL_ALGO', 'NCCL_PROTO', 'HSA_ENABLE_IPC_MODE_LEGACY')},
                                 'rccl_version': rccl_version(),
                                 'rank_ms_per_step': {'min': round(rank_time['min_s'] / args.steps * 1e3, 3),
                                                      'max': round(rank_time['max_s'] / args.steps * 1e3, 3)}}
            if ddp_stats:
                ex = sorted(st['exposed_ms'] for st in ddp_stats if 'exposed_ms' in st)
                last = ddp_stats[-1]
                out['collective']['exposed_ms'] = {
                    'median': ex[len(ex) // 2], 'max': ex[-1], 'steps': len(ex),
                    'what': 'rank 0, main stream: from the end of its last backward kernel to the end of the last all-reduce '
                            '(events around GradAllReducer.finish()) = what the collectives add behind backward'} if ex else None
                out['collective']['buckets'] = {'bytes': last['bucket_bytes'], 'launch_offset_ms': last['launch_offset_ms'],
                                                'backward_ms': last.get('backward_ms'),
                                                'what': 'launch = the point on the producing stream (wgrad / side) behind which the '
                                                        'bucket\'s all-reduce is ordered, ms after the start of backward; last '
                                                        'profiled step'}
        if world == 1 and not args.no_cpu_baseline:
            # the thread count is swept first (one warm-up + two steps per point, the faithful variant), the 3 + 5 protocol then runs
            # at the best point: BASELINE.md asks for the host's best, and torch CPU has an optimum well below the logical core count
            sweep = orc.sweep_cpu_threads((16, 32, 64), variant='faithful')
            best = next((d['threads'] for d in sweep if 's_per_step' in d), 16)
            variants = {}
            for name in ('faithful', 'label_map'):
                v, cores, sample = orc.time_cpu_baseline(iters=5, warmup=3, variant=name, threads=best)
                sample += f'; {cores} threads = the best of the sweep, {os.cpu_count()} logical cores available'
                variants[name] = {'value': round(v, 4), 'unit': 'images/s', 'cores': cores, 'sample': sample}
            f = variants['faithful']
            out['cpu_baseline'] = {'value': f['value'], 'unit': 'images/s', 'cores': f['cores'], 'kind': 'port',
                                   'sample': f['sample'], 'host': host_cpu(), 'sweep': sweep,
                                   'what': "oracle/wesup_oracle.py, variant 'faithful' = the reference's own algorithm "
                                           '(dense sp_maps, incremental cat, dense mm; models/wesup.py:18-63,246-304,492-531 '
                                           "+ backward); 'label_map' = the scatter-mean restatement of the same step",
                                   'variants': variants}
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, argv))
    worker(args)


if __name__ == '__main__':
    main()
