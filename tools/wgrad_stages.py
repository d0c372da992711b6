"""The kernels of one Winograd-domain weight gradient per layer, alone on the GPU, from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wg -- python3 tools/wgrad_stages.py run
  python3 tools/wgrad_stages.py report gpurun_out/wg"""
import sys, os, glob, csv, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == 'run':
    import torch
    from wesup_amd import ops
    from wesup_amd.engine import CONV_CH, POOL_AFTER
    d = torch.device('cuda:0')
    B, H, W, m = 4, 480, 480, 4
    marker = torch.zeros(64, device=d)
    h, w = H, W
    for l, (ci, co) in enumerate(CONV_CH):
        if l >= 1:
            x = torch.relu(torch.randn(B, h, w, ci, device=d))
            dy = torch.randn(B, h, w, co, device=d)
            V = ops.winograd_input_transform(x, m=m)
            dw = torch.empty(co, ci, 3, 3, device=d); db = torch.empty(co, device=d)
            for _ in range(3):
                ops.conv3x3_wgrad_winograd(x, dy, relu_in=False, dw=dw, db=db, v_pre=V, m=m)
            torch.cuda.synchronize()
            marker.fill_(float(l))
            ops.conv3x3_wgrad_winograd(x, dy, relu_in=False, dw=dw, db=db, v_pre=V, m=m)
            torch.cuda.synchronize()
            del x, dy, V
        if POOL_AFTER[l]:
            h, w = h // 2, w // 2
else:
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    layer, on = 0, False
    for r in rows:
        n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        if 'FillFunctor' in n and int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) <= 256:
            layer += 1; on = True; tot = 0.0
            print(f'--- layer {layer}')
            continue
        if on:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            print(f'   {d:8.1f} us  grid {r.get("Grid_Size_X", "?"):>8} x {r.get("Grid_Size_Y", "?"):>4} x {r.get("Grid_Size_Z", "?"):>4}  {n[:70]}')
            if 'wino_wgrad_reduce' in n or 'colsum_stage2' in n and False:
                on = 'bias' if False else on
            if 'wino_wgrad_reduce' in n:
                on = False
