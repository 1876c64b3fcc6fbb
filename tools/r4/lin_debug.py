import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cstp_amd import ops
for (b, fin, fout) in [(16, 4096, 512), (32, 4096, 512), (32, 512, 4096), (24, 1024, 1000)]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand((b, fin), generator=g, dtype=torch.float64) * 2 - 1
    w = (torch.rand((fout, fin), generator=g, dtype=torch.float64) * 2 - 1) * 0.1
    y = x @ w.t()
    yg = ops.linear(x.float().cuda(), w.float().cuda(), None).double().cpu()
    err = (yg - y).abs()
    print(b, fin, fout, "max err", float(err.max()), "ref max", float(y.abs().max()))
    print("  per-row max err", [round(float(v), 4) for v in err.amax(1)])
    bad = (err > 1e-3).nonzero()
    print("  bad count", bad.shape[0], "first", bad[:5].tolist(), "k range", (int(bad[:, 1].min()), int(bad[:, 1].max())) if bad.shape[0] else None)
    # which slice is missing?  y - yg should equal the contribution of one slice
    if bad.shape[0]:
        n, k = bad[0].tolist()
        d = float(y[n, k] - yg[n, k])
        contrib = [(float((x[n, s * 64:(s + 1) * 64] * w[k, s * 64:(s + 1) * 64]).sum()), s) for s in range(fin // 64)]
        best = min(contrib, key=lambda t: abs(t[0] - d))
        print("  diff", d, "closest slice contribution", best)
