import sys, torch
sys.path.insert(0, "/root/repo")
from cstp_amd import ops
ops.set_split_terms(2)
xs = (1, 16, 8, 14, 14); k = 16
ws = (k, 16, 3, 1, 1)
ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, 4, 0, 0))
x = torch.zeros(xs)
pos = torch.arange(8 * 196, dtype=torch.float32).reshape(8, 14, 14)
x[0, 0] = pos
x[0, 1] = 10000 + pos
w = torch.zeros(ws)
w[0, 0, 1] = 1.0   # y[0] = x[c0] (centre tap)
w[1, 1, 1] = 1.0   # y[1] = x[c1]
w[2, 0, 0] = 1.0   # y[2] = x[c0] shifted: tap 0 -> frame d-1
w[3, 0, 2] = 1.0
y = ops.conv3d(x.cuda(), w.cuda(), None, 1, (1, 0, 0)).cpu()
print("y0 first 32:", y[0, 0].flatten()[:32].tolist())
print("y1 first 8:", y[0, 1].flatten()[:8].tolist())
print("y2 d=1 first 8:", y[0, 2, 1].flatten()[:8].tolist(), "expect", pos[0].flatten()[:8].tolist())
print("y3 d=0 first 8:", y[0, 3, 0].flatten()[:8].tolist(), "expect", pos[1].flatten()[:8].tolist())
ref = torch.nn.functional.conv3d(x, w, None, 1, (1, 0, 0))
bad = (y - ref).abs() > 1e-3
print("bad count", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()[:10]
for i in idx: print(i.tolist(), float(y[tuple(i)]), float(ref[tuple(i)]))
