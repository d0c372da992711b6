"""Kernel-by-kernel durations of ONE training step (single-stream schedule) from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/step_trace.py run [multi]
  python3 tools/step_trace.py report gpurun_out/trace > gpurun_out/step_trace.txt
On the GPU box (one gpurun call; `multi` = the three-stream schedule, WESUP_TRACE_BATCH = images per step, WESUP_TRACE_SKIP =
launch classes left out, timing only):
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export GPU_MAX_HW_QUEUES=6 WESUP_TRACE_BATCH=1 &&
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/step_trace.py run multi > gpurun_out/trace.log 2>&1 &&
  python3 tools/step_trace.py report gpurun_out/trace > gpurun_out/step_trace_b1.txt
(A caution, DESIGN.md 6 round 4: the tracer stretches cross-queue waits of short kernels; compare gaps with untraced A/B runs.)
"""
import sys, os, glob, csv, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == 'run':
    import torch
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    dev = torch.device('cuda:0')
    B, H, W, g = int(os.environ.get('WESUP_TRACE_BATCH', '4')), 480, 480, 24
    trainer = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in orc.make_weights(0, feat_scale=0.05).items()})
    trainer.optimizer, _ = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.tracker.train()
    trainer.model.engine.two_streams = len(sys.argv) > 2 and sys.argv[2] == 'multi'
    trainer.model.engine._diag_skip = set(filter(None, os.environ.get('WESUP_TRACE_SKIP', '').split(',')))      # timing-only
    for kv in filter(None, os.environ.get('WESUP_TRACE_ENGINE_SET', '').split(',')):      # e.g. plain=1
        k, v = kv.split('=')
        setattr(trainer.model.engine, k, type(getattr(trainer.model.engine, k))(int(v)))
    imgs, labs, pts, pix = synth.make_batch(1, B, H, W, g)
    data = (torch.from_numpy(imgs).to(dev), torch.from_numpy(pix).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(labs).to(dev))
    for _ in range(8):          # (the last iterations replay the recorded step plan: the host is out of the picture)
        trainer.train_one_iteration('train', *data)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # last step = kernels after the last-but-one sgd_kernel
    sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
    # (round 5: the optimiser step is two launches a few kernels apart -- a step ends with the LAST sgd_kernel of such a group)
    ends = [i for k, i in enumerate(sgd) if k + 1 == len(sgd) or sgd[k + 1] - i > 40]
    lo = ends[-2] + 1
    step = rows[lo:ends[-1] + 1]
    t0 = int(step[0]['Start_Timestamp'])
    agg = {}
    for r in step:
        n = r['Kernel_Name']
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {d:9.1f} us  q{r.get('Queue_Id', '?'):>2} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {n}")
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += d
    print('--- totals')
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{d:10.1f} us {c:4d}  {n}')
    print('sum of kernel durations %.1f us, span %.1f us' % (sum(v[1] for v in agg.values()), (int(step[-1]['End_Timestamp']) - t0) / 1e3))
    # ---- concurrency view: how long is any / an MFMA kernel in flight, how much do kernels overlap
    ev = []
    for r in step:
        n = r['Kernel_Name']
        mf = ('gemm_nt_kernel' in n) or ('gemm_tn_kernel' in n) or ('wino4_gemm_out_kernel' in n)
        ev.append((int(r['Start_Timestamp']), 1, mf))
        ev.append((int(r['End_Timestamp']), -1, mf))
    ev.sort()
    any_n = mf_n = 0
    last = ev[0][0]
    t_any = t_mf = t_mf2 = t_idle = 0
    for t, d, mf in ev:
        dt = t - last
        if any_n > 0: t_any += dt
        else: t_idle += dt
        if mf_n > 0: t_mf += dt
        if mf_n > 1: t_mf2 += dt
        last = t
        any_n += d
        if mf: mf_n += d
    print('in flight: any kernel %.1f us, idle %.1f us, >=1 MFMA GEMM %.1f us, >=2 MFMA GEMMs %.1f us' % (t_any / 1e3, t_idle / 1e3, t_mf / 1e3, t_mf2 / 1e3))
    # ---- intervals without any MFMA GEMM in flight: what runs there
    gaps = []
    mf_n = 0
    last = None
    for t, d, mf in ev:
        if mf:
            if mf_n == 0 and d == 1 and last is not None and t - last > 3000:
                gaps.append((last, t))
            mf_n += d
            if mf_n == 0:
                last = t
        elif last is None and mf_n == 0:
            last = ev[0][0]
    gaps.sort(key=lambda g: g[0] - g[1])
    print('--- longest intervals with no MFMA GEMM in flight (us from step start, length, kernels running inside)')
    for a, b in sorted(gaps[:14]):
        inside = {}
        for r in step:
            s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if e_ > a and s_ < b:
                n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:40]
                inside[n] = inside.get(n, 0) + (min(e_, b) - max(s_, a)) / 1e3
        top = sorted(inside.items(), key=lambda kv: -kv[1])[:4]
        print(f'{(a - t0) / 1e3:9.1f} {(b - a) / 1e3:7.1f} us  ' + ', '.join(f'{k} {v:.0f}' for k, v in top))
    print('total no-GEMM time in intervals > 3 us: %.1f us' % (sum(b - a for a, b in gaps) / 1e3))
