import sys, itertools, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import winograd_oracle as wo
import torch

def run(points, m, x, w, dtype=np.float32):
    AT, G, BT = wo.toom_cook(points, m)
    n = m + 2
    B, H, W, C = x.shape
    Th, Tw = (H + m - 1)//m, (W + m - 1)//m
    xp = np.zeros((B, m*Th+2, m*Tw+2, C), dtype=dtype); xp[:, 1:H+1, 1:W+1] = x
    d = wo._split(xp, m, Th, Tw, n, m, 1)
    bt = BT.astype(dtype); g = G.astype(dtype); at = AT.astype(dtype)
    # two-stage separable transform in fp32 (like a kernel would)
    t = np.einsum('ar,bijrcC->bijacC', bt, d).astype(dtype)
    v = np.einsum('bijacC,dc->bijadC', t, bt).astype(dtype)
    V = v.reshape(B*Th*Tw, n*n, C).transpose(1, 0, 2).copy()
    w64 = w.astype(np.float64)
    uf = np.einsum('ar,oirc,dc->adoi', G, w64, G).reshape(n*n, w.shape[0], w.shape[1]).astype(dtype)
    M = np.matmul(V, uf.transpose(0, 2, 1)).astype(dtype)
    mm = M.transpose(1, 0, 2).reshape(B, Th, Tw, n, n, -1)
    t = np.einsum('ar,bijrcC->bijacC', at, mm).astype(dtype)
    y = np.einsum('bijacC,dc->bijadC', t, at).astype(dtype)
    y = y.transpose(0, 1, 3, 2, 4, 5).reshape(B, m*Th, m*Tw, -1)[:, :H, :W]
    return y

def ref(x, w):
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w.astype(np.float64))
    return torch.nn.functional.conv2d(xt, wt, padding=1).permute(0, 2, 3, 1).numpy()

rng = np.random.RandomState(0)
C = 512; H = 30
x = np.maximum(rng.randn(1, H, H, C), 0).astype(np.float32)
w = (rng.randn(C, C, 3, 3) * np.sqrt(2.0/(9*C))).astype(np.float32)
yr = ref(x, w); mx = np.abs(yr).max()
# direct fp32 yardstick
xt = torch.from_numpy(x).permute(0,3,1,2); wt = torch.from_numpy(w)
yd = torch.nn.functional.conv2d(xt, wt, padding=1).permute(0,2,3,1).numpy()
print('direct fp32', np.abs(yd - yr).max()/mx)
y4 = run(wo.F4_POINTS, 4, x, w)
print('F4 tuned', np.abs(y4 - yr).max()/mx)
y4 = run((0,1,-1,2,-2), 4, x, w)
print('F4 textbook', np.abs(y4 - yr).max()/mx)
std = (0, 1, -1, 2, -2, (1,2), (-1,2))
print('F6 textbook', np.abs(run(std, 6, x, w) - yr).max()/mx)
cands = [(1,4),(3,8),(1,2),(5,8),(3,4),(7,8),(1,1),(9,8),(5,4),(3,2),(7,4),(2,1),(5,2),(3,1)]
res = []
for a, b, c in itertools.combinations(cands, 3):
    pts = (0, a, (-a[0], a[1]), b, (-b[0], b[1]), c, (-c[0], c[1]))
    try:
        e = np.abs(run(pts, 6, x, w) - yr).max()/mx
    except Exception as ex:
        continue
    res.append((e, a, b, c))
res.sort()
for r in res[:15]: print(r)
