"""Cost of one in-place all-reduce of the flat gradient (75.5 MB fp32) on the RCCL stream, single rank: the fixed
latency the data-parallel step pays between the last gradient and the SGD kernel."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
for n in (18868194, 4 << 20, 1 << 20, 1 << 16):
    x = torch.ones(n, device='cuda')
    for _ in range(3):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(10):
        w = dist.all_reduce(x, async_op=True); w.wait()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'all_reduce {n*4/1e6:8.2f} MB: {e0.elapsed_time(e1)/10*1e3:8.1f} us per call on the GPU, {1e6*(t1-t0)/10:8.1f} us of host time per call')
dist.destroy_process_group()
