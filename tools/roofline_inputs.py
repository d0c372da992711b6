"""Builds profiles/rNN_roofline_inputs.json and profiles/rNN_layer_mfma.csv from the rocprofv3 --pmc passes that
tools/collect_profiles.sh ran.  bench.py reads the JSON for `roofline.traffic`; nothing in it is typed by hand.

  python tools/roofline_inputs.py OUT_DIR TAG [BATCH SIZE GRID] > summary      (TAG: r03, r03_c4, ...; default shape 4 480 24)

Inputs (all under OUT_DIR):
  pmc_bench_fetch/, pmc_bench_write/, pmc_bench_mfma/, pmc_bench_busy/ : counter passes over `bench.py` itself
      (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES; GRBM_GUI_ACTIVE -- passes of their own, MI355X_MICROARCH.md)
  pmc_layer_mfma/, pmc_layer_busy/ + layer_manifest.json : the same two MFMA counters over tools/layer_pmc.py

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB, summed over the
XCDs; gfx950 FETCH_SIZE under-reports wide streaming reads 2x, hence HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)  (ROCm 7.2 has no gfx950
derived-counter section, so the ratio is formed by hand).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def read_pass(d):
    """[(dispatch_id, kernel, counter, value)] with the per-XCD / per-SE rows of a dispatch summed."""
    acc = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (int(r['Dispatch_Id']), r['Kernel_Name'], r['Counter_Name'])
            acc[k] = acc.get(k, 0.0) + float(r['Counter_Value'])
    return [(k[0], k[1], k[2], v) for k, v in sorted(acc.items())]


def klass(kernel):
    """conv3x3 fwd+dgrad GEMMs | conv3x3 wgrad GEMMs | scatter_mean | None.  gemm_nt_kernel MODE 1/2 = implicit-GEMM conv,
    MODE 3 = the batched Winograd-domain products (MODE 0: side convs / MLP, not counted here); gemm_tn_kernel likewise."""
    if 'wino4_gemm_out_kernel' in kernel:      # the batched products + output transform of the short products in one kernel
        return 'conv3x3_fwd_dgrad'
    m = re.search(r'gemm_nt_kernel<([^>]*)>', kernel)
    if m:
        mode = int(m.group(1).split(',')[5])
        return 'conv3x3_fwd_dgrad' if mode in (1, 2, 3) else None
    m = re.search(r'gemm_tn_kernel<([^>]*)>', kernel)
    if m:
        mode = int(m.group(1).split(',')[4])
        return 'conv3x3_wgrad' if mode in (1, 2, 3) else None
    if kernel.startswith('sp_pool_fwd_kernel'):
        return 'scatter_mean'
    return None


def per_class(rows, counter):
    tot, n = collections.Counter(), collections.Counter()
    for _, kern, ctr, v in rows:
        c = klass(kern)
        if c and ctr == counter:
            tot[c] += v
            n[c] += 1
    return tot, n


def main():
    out_dir, rnd = sys.argv[1], sys.argv[2]
    B, H, g = (int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else (4, 480, 24)
    W, N = H, g * g
    res = {'round': rnd, 'shape': {'batch': B, 'H': H, 'W': W, 'superpixels': N},
           'how': f'rocprofv3 --pmc passes over `python3 bench.py --batch {B} --size {H} --grid {g} --steps 2 --warmup 1 --no-cpu-baseline` (one counter '
                  'group per pass, no trace domain beside it); per-launch means over every launch of the kernel class in '
                  'the run; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)'}
    fetch, nf = per_class(read_pass(os.path.join(out_dir, 'pmc_bench_fetch')), 'FETCH_SIZE')
    write, nw = per_class(read_pass(os.path.join(out_dir, 'pmc_bench_write')), 'WRITE_SIZE')
    mfma, nm = per_class(read_pass(os.path.join(out_dir, 'pmc_bench_mfma')), 'SQ_VALU_MFMA_BUSY_CYCLES')
    busy, nb = per_class(read_pass(os.path.join(out_dir, 'pmc_bench_busy')), 'GRBM_GUI_ACTIVE')
    # algorithmic bytes of the implicit GEMMs per step at this shape: read x once + write y once (+ weights), per layer
    from_layers = conv_algorithmic_bytes(B, H, W)
    res['memory_bound'] = memory_bound_classes(out_dir)
    for c in ('conv3x3_fwd_dgrad', 'conv3x3_wgrad', 'scatter_mean'):
        e = {}
        if nf[c] and nw[c]:
            e['launches_counted'] = nf[c]
            e['fetch_kib_per_launch'] = fetch[c] / nf[c]
            e['write_kib_per_launch'] = write[c] / nw[c]
            e['hbm_bytes_per_launch'] = (2.0 * fetch[c] / nf[c] + write[c] / nw[c]) * 1024.0
        if nm[c] and nb[c]:
            e['mfma_busy_frac'] = mfma[c] / (1024.0 * busy[c] / 8.0) if busy[c] else None
        if c in from_layers:
            e['algorithmic_bytes_per_launch'] = from_layers[c]
        res[c] = e
    res['scatter_mean']['algorithmic_bytes_per_launch'] = 4.0 * B * (2112 * H * W + H * W + N * 2112)
    with open(os.path.join('profiles', f'{rnd}_roofline_inputs.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))

    # ---- per-layer table: 13 layers x {fwd, dgrad, wgrad}: MFMA pipe busy, FLOP-weighted summary
    man_path = os.path.join(out_dir, 'layer_manifest.json')
    if not os.path.exists(man_path):
        return
    man = json.load(open(man_path))
    reps = man['reps']

    def main_launches(d, counter):
        rows = [(i, k, v) for i, k, c, v in read_pass(d) if c == counter and klass(k) in ('conv3x3_fwd_dgrad', 'conv3x3_wgrad')]
        return [v for _, _, v in sorted(rows)]
    m = main_launches(os.path.join(out_dir, 'pmc_layer_mfma'), 'SQ_VALU_MFMA_BUSY_CYCLES')
    g = main_launches(os.path.join(out_dir, 'pmc_layer_busy'), 'GRBM_GUI_ACTIVE')
    assert len(m) == len(g) == reps * len(man['order']), (len(m), len(g), reps * len(man['order']))
    lines = ['layer,pass,gflop,mfma_busy_cycles,kernel_cycles,mfma_pipe_busy_frac,tflops_at_counter_clock']
    wsum = collections.Counter()
    wfl = collections.Counter()
    for gi, (layer, pas, fl) in enumerate(man['order']):
        mm = sum(m[gi * reps + 1:(gi + 1) * reps]) / (reps - 1)          # first launch of a group = warm-up
        cyc = sum(g[gi * reps + 1:(gi + 1) * reps]) / (reps - 1) / 8.0
        frac = mm / (1024.0 * cyc)
        lines.append(f'{layer},{pas},{fl / 1e9:.2f},{mm:.0f},{cyc:.0f},{frac:.4f},')
        wsum[pas] += frac * fl
        wfl[pas] += fl
    for pas in ('fwd', 'dgrad', 'wgrad'):
        lines.append(f'all,{pas},{wfl[pas] / 1e9:.2f},,,{wsum[pas] / wfl[pas]:.4f},  # FLOP-weighted')
    tot = sum(wsum.values()) / sum(wfl.values())
    lines.append(f'all,all,{sum(wfl.values()) / 1e9:.2f},,,{tot:.4f},  # FLOP-weighted over 13 layers x 3 passes')
    with open(os.path.join('profiles', f'{rnd}_layer_mfma.csv'), 'w') as f:
        f.write(f'# MFMA pipe busy per conv3x3 layer and pass at B={man["shape"][0]}, {man["shape"][1]}x{man["shape"][2]}, kernels alone on the GPU (the engine\'s routing):\n'
                '# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8), rocprofv3 --pmc over tools/layer_pmc.py (two passes),\n'
                '# mean of the launches after the first of each group; the main GEMM launch only (stream-K fix-up / split-K reduce\n'
                '# launches carry no MFMA and are listed in the kernel-stats file).  SURVEY.md 8(d).\n')
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


MEM_KERNELS = {      # kernel-name prefix -> class of the memory-bound report (per-launch counter traffic next to the event times)
    'sgd_kernel': 'sgd', 'prop_kernel': 'propagate', 'paint_kernel': 'paint', 'sp_hist_kernel': 'sp_preprocess',
    'sp_order_kernel': 'sp_preprocess', 'sp_chunk_base_kernel': 'sp_preprocess', 'sp_place_kernel': 'sp_preprocess',
    'sp_segments_kernel': 'sp_preprocess', 'sp_count_order_kernel': 'sp_preprocess', 'sp_pool_up_fwd_kernel': 'sp_pool_up_fwd', 'sp_interp_matrix_kernel': 'interp_matrix',
    'wino4_input_transform_kernel': 'winograd_transform', 'wino4_output_transform_kernel': 'winograd_transform',
    'wino_input_transform_kernel': 'winograd_transform', 'wino_output_transform_kernel': 'winograd_transform',
    'wino4_dual_transform_kernel': 'winograd_transform', 'wino4_outgrad_transform_kernel': 'winograd_outgrad_transform', 'upsample_bwd_cell_kernel': 'upsample_bwd',
    'maxpool_bwd_kernel': 'maxpool_bwd'}


def memory_bound_classes(out_dir):
    """Counter HBM bytes per LAUNCH ((2*FETCH + WRITE) * 1024) of the memory-bound kernel classes in the bench passes."""
    fetch = read_pass(os.path.join(out_dir, 'pmc_bench_fetch'))
    write = read_pass(os.path.join(out_dir, 'pmc_bench_write'))
    acc = {}
    for rows, key in ((fetch, 'fetch_kib'), (write, 'write_kib')):
        for _, kern, ctr, v in rows:
            name = re.sub(r'^void ', '', kern).split('(')[0].split('<')[0]
            c = MEM_KERNELS.get(name)
            if c:
                e = acc.setdefault(c, {'fetch_kib': 0.0, 'write_kib': 0.0, 'n_fetch_kib': 0, 'n_write_kib': 0})
                e[key] += v
                e['n_' + key] += 1
    out = {}
    for c, e in acc.items():
        if e['n_fetch_kib'] and e['n_write_kib']:
            out[c] = {'launches_counted': e['n_fetch_kib'],
                      'hbm_bytes_per_launch': (2.0 * e['fetch_kib'] / e['n_fetch_kib'] + e['write_kib'] / e['n_write_kib']) * 1024.0}
    return out


def conv_algorithmic_bytes(B, H, W, wino_min_ci=64, m=4):
    """Mean algorithmic HBM bytes per GEMM launch of the two conv kernel classes over one step: every operand read once,
    every result written once.  Direct layers (below wino_min_ci input channels) -- fwd: x, w, y; dgrad: dy, w, mask +
    old dx + new dx; wgrad: x, dy, dw.  Winograd-domain layers (F(m x m,3x3), P = (m+2)^2 positions) -- the GEMM launch
    reads the transformed input (P x tiles x Cin), the transformed filter (P x Cin x Cout) and writes the transformed
    output (P x tiles x Cout), or, where the product is short (K <= 256: wino4_gemm_out_kernel, products + output transform
    in one kernel), the output itself (forward: y; input gradient: mask + old + new dx); the wgrad launch reads both
    transformed tensors and writes P filter-gradient slabs."""
    ch = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512),
          (512, 512), (512, 512), (512, 512)]
    pool = [False, True, False, True, False, False, True, False, False, True, False, False, False]
    h, w = H, W
    nt, tn, n_nt, n_tn = 0.0, 0.0, 0, 0
    for l, (ci, co) in enumerate(ch):
        px = B * h * w
        cin = 4 if l == 0 else ci
        if ci >= wino_min_ci:
            T, P = B * ((h + m - 1) // m) * ((w + m - 1) // m), (m + 2) ** 2
            for k, n, outs in ((ci, co, 1.0), (co, ci, 3.0)):         # fwd (K = ci: writes y), dgrad (K = co: mask + old + new dx)
                if m == 4 and k <= 256 and n % 64 == 0:               # fused products + output transform
                    nt += 4.0 * (P * T * k + P * k * n + outs * px * n)
                else:
                    nt += 4.0 * P * (T * k + k * n + T * n)
            n_nt += 2
            tn += 4.0 * P * (T * ci + T * co + ci * co)
            n_tn += 1
        else:
            nt += 4.0 * (px * cin + 9 * cin * co + px * co)
            n_nt += 1
            if l > 0:
                nt += 4.0 * (px * co + 9 * ci * co + 3 * px * ci)
                n_nt += 1
            tn += 4.0 * (px * cin + px * co + 9 * ci * co)
            n_tn += 1
        if pool[l]:
            h, w = h // 2, w // 2
    return {'conv3x3_fwd_dgrad': nt / n_nt, 'conv3x3_wgrad': tn / n_tn}


if __name__ == '__main__':
    main()
