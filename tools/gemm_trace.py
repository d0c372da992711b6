# needs the debug library (the shipped kernels carry no clock / trace code):
#   make -C wesup_amd/csrc debug && WESUP_HIP_LIB=wesup_amd/csrc/libwesup_hip_debug.so python tools/gemm_trace.py
"""Per-block timeline of one NT GEMM / conv launch (debug trace in gemm.hip): where does a tile spend its time?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wesup_amd import ops, _lib
ops.set_streamk(everything=True)
d = torch.device('cuda:0')
lib = _lib.load()
def run(name, fn, nblocks):
    buf = torch.zeros(nblocks * 6, dtype=torch.int64, device=d)
    fn(); torch.cuda.synchronize()
    lib.wesup_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.wesup_debug_set_trace(None)
    raw = buf.cpu().numpy().reshape(nblocks, 6)
    t = raw[:, :4].astype(np.float64) * 0.01    # us
    xcc = raw[:, 4] & 0xf; hw = raw[:, 5]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    t0 = t[:, 0].min()
    st, ls, le, en = (t[:, i] - t0 for i in range(4))
    print(f'{name}: blocks {nblocks}  kernel span {en.max():.1f} us')
    print(f'   block start     : p0 {st.min():.1f} p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}')
    print(f'   prologue (start->loop) p50 {np.median(ls - st):.2f} max {(ls - st).max():.2f}')
    print(f'   main loop       : p10 {np.percentile(le - ls, 10):.1f} p50 {np.median(le - ls):.1f} p90 {np.percentile(le - ls, 90):.1f} max {(le - ls).max():.1f}')
    print(f'   epilogue        : p50 {np.median(en - le):.2f} max {(en - le).max():.2f}')
    print(f'   block end       : p10 {np.percentile(en, 10):.1f} p50 {np.median(en):.1f} p90 {np.percentile(en, 90):.1f} max {en.max():.1f}')
    loop = le - ls
    print('   loop p50 by XCC:', {int(x): round(float(np.median(loop[(xcc == x) & (st < 5)])), 1) for x in sorted(set(xcc.tolist()))})
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    firstw = st < 5
    groups = {}
    for k_, l_ in zip(key[firstw].tolist(), loop[firstw].tolist()): groups.setdefault(k_, []).append(l_)
    sizes = [len(v) for v in groups.values()]
    pair_diff = [abs(v[0] - v[1]) for v in groups.values() if len(v) == 2]
    print(f'   distinct CUs {len(groups)}, blocks/CU histogram {np.bincount(sizes).tolist()}, same-CU pair |dt| p50 {np.median(pair_diff) if pair_diff else -1:.1f} us')
    cu_mean = np.array([np.mean(v) for v in groups.values()])
    print(f'   per-CU mean loop: p10 {np.percentile(cu_mean,10):.1f} p50 {np.median(cu_mean):.1f} p90 {np.percentile(cu_mean,90):.1f}; by blocks/CU:', {n: round(float(np.mean([np.mean(v) for v in groups.values() if len(v) == n])), 1) for n in sorted(set(sizes))})
    if nblocks > 512:      # blocks behind the whole tiles are stream-K parts
        pt = np.arange(nblocks) >= 512
        print(f'   whole tiles: start p50 {np.median(st[~pt]):.1f}, end p10 {np.percentile(en[~pt],10):.1f} p50 {np.median(en[~pt]):.1f} p90 {np.percentile(en[~pt],90):.1f} max {en[~pt].max():.1f}')
        print(f'   parts      : start p10 {np.percentile(st[pt],10):.1f} p50 {np.median(st[pt]):.1f} p90 {np.percentile(st[pt],90):.1f}, end p10 {np.percentile(en[pt],10):.1f} p50 {np.median(en[pt]):.1f} p90 {np.percentile(en[pt],90):.1f} max {en[pt].max():.1f}; duration p10 {np.percentile((en-st)[pt],10):.1f} p50 {np.median((en-st)[pt]):.1f} p90 {np.percentile((en-st)[pt],90):.1f}')
    first = st < 1.0
    print(f'   first-wave blocks: {first.sum()}  their loop p50 {np.median((le - ls)[first]):.1f}; later blocks loop p50 {np.median((le - ls)[~first]) if (~first).any() else 0:.1f}')
case = sys.argv[1] if len(sys.argv) > 1 else 'wide'
if case == 'wide':
    M, N, K = 32768, 256, 2304
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
    run('gemm 512 tiles K=2304', lambda: ops.gemm_nt(A, B, None, out=C), 512)
    x = torch.randn(4, 120, 120, 256, device=d); w = torch.randn(256, 256, 3, 3, device=d) * 0.02
    wf, _ = ops.pack_conv3x3_weight(w, need_dgrad=False); y = torch.empty(4, 120, 120, 256, device=d); bias = torch.zeros(256, device=d)
    run('conv fwd L6 (900 tiles: 512 whole + 512 stream-K parts)', lambda: ops.conv3x3_fwd(x, wf, bias, 256, True, out=y), 1024)
else:
    # the 64-channel layers: conv1_2 (64 -> 64 at 480x480: 7200 tiles of 128x64, 18 K-steps each) forward and dgrad,
    # conv2_1 (64 -> 128 at 240x240: 1800 tiles of 128x128, 18 K-steps)
    x = torch.randn(4, 480, 480, 64, device=d); w = torch.randn(64, 64, 3, 3, device=d) * 0.05
    wf, wd = ops.pack_conv3x3_weight(w); y = torch.empty(4, 480, 480, 64, device=d); bias = torch.zeros(64, device=d)
    run('conv1_2 fwd (7200 tiles 128x64, K = 576)', lambda: ops.conv3x3_fwd(x, wf, bias, 64, True, out=y), 7200)
    dx = torch.zeros(4, 480, 480, 64, device=d)
    run('conv1_2 dgrad (mask + accumulate)', lambda: ops.conv3x3_dgrad(y, wd, 64, mask_src=x, out=dx, accumulate=True), 7200)
    x2 = torch.randn(4, 240, 240, 64, device=d); w2 = torch.randn(128, 64, 3, 3, device=d) * 0.05
    wf2, _ = ops.pack_conv3x3_weight(w2, need_dgrad=False); y2 = torch.empty(4, 240, 240, 128, device=d); b2 = torch.zeros(128, device=d)
    run('conv2_1 fwd (1800 tiles 128x128, K = 576)', lambda: ops.conv3x3_fwd(x2, wf2, b2, 128, True, out=y2), 2048)
