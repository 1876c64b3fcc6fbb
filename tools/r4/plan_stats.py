import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cstp_amd.optim import FlatSGD
from cstp_amd.r21d_byol import R21DBYOL, layer_sizes_for_depth
from cstp_amd.synthetic import device_batch
from cstp_amd.train import PretrainStep
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 34
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(depth)).cuda()
arenas = model.flatten_parameters(); model.train()
opt = FlatSGD(model.parameters(), lr=0.09, momentum=0.9, weight_decay=5e-4, arenas=arenas)
step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 0.0, 0.0), clip_grad_norm=True)
x1, x2, lab = device_batch(16, 16, 112, dev, seed=1)
for i in range(8):
    step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
    p = step._packs
    print(i, p.state, p.stats, len(p.recs), {k: (v[2], v[3]) for k, v in p.tables.items()}, sum(w.numel() for w in p.ws.values()) / 2**20, "MiB")
