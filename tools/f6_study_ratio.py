import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/tmp/f6')
import importlib.util
src = open('/tmp/f6/study.py').read().split("rng = np.random.RandomState(0)")[0]
exec(src)
from oracle import winograd_oracle as wo
for C, H, relu in ((512, 30, True), (512, 30, False), (512, 60, True), (256, 60, False)):
    rng = np.random.RandomState(1)
    x = rng.randn(1, H, H, C); x = np.maximum(x, 0) if relu else x
    x = x.astype(np.float32)
    w = (rng.randn(C, C, 3, 3) * np.sqrt(2.0/(9*C))).astype(np.float32)
    yr = ref(x, w); mx = np.abs(yr).max()
    e4 = np.abs(run(wo.F4_POINTS, 4, x, w) - yr).max()/mx
    res = {}
    for name, pts in (('textbook', (0,1,-1,2,-2,(1,2),(-1,2))), ('5/8,1,7/4', (0,(5,8),(-5,8),1,-1,(7,4),(-7,4))), ('1/2,1,2', (0,(1,2),(-1,2),1,-1,2,-2))):
        res[name] = np.abs(run(pts, 6, x, w) - yr).max()/mx
    print(f'C={C} H={H} relu={relu}: F4 tuned {e4:.2e}; F6 ' + ', '.join(f'{k} {v:.2e} ({v/e4:.1f}x)' for k, v in res.items()))
