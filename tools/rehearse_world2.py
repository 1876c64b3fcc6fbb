#!/usr/bin/env python3
"""Rehearse the N > 1 control flow of the pre-training step with the REAL HIP kernels on a one-GPU box: two ranks, both on
cuda:0, process group over gloo (RCCL refuses two ranks on one device).  Everything of the multi-rank step runs except RCCL
itself and the NT-Xent all-gather (gloo has no CUDA all_gather): tile-table broadcast, BN-buffer broadcast at the top of every
step, forward/backward under no_sync(), the gradient arena all-reduced in six slices started from backward hooks
(train.StagedAllReduce), clip, SGD, lagged log all-reduce.
Checks after each step: parameters and target parameters bit-identical across the ranks (the BN running statistics are per rank
between two broadcasts, as under the reference's DDP); after the last step: sync_buffers (what validation / checkpointing call)
makes the buffers identical too.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/rehearse_world2.py [depth] [B_local] [steps]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
b_local = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)

from cstp_amd import ops  # noqa: E402
from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r21d_byol import R21DBYOL, layer_sizes_for_depth  # noqa: E402
from cstp_amd.synthetic import device_batch  # noqa: E402
from cstp_amd.train import LaggedScalars, PretrainStep  # noqa: E402

torch.manual_seed(1)
model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(depth)).cuda(0)
arenas = model.flatten_parameters()
model.train()
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0)
opt = FlatSGD(model.parameters(), lr=0.09, momentum=0.9, weight_decay=5e-4, arenas=arenas)
step = PretrainStep(ddp, opt, (0.1, 1.0, 1.0, 1.0, 1.0), clip_grad_norm=True)
x1, x2, lab = device_batch(b_local, 16, 112, dev, seed=1 + rank)          # a different shard per rank
lagged = LaggedScalars(dev, world)
if rank == 1:                                                               # rank 1's BN statistics start out different:
    with torch.no_grad():                                                   # the per-step broadcast must overwrite them
        arenas["buffers"].add_(1.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(steps):
    if it == 1:
        ev[0].record()
    out = step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
    late = lagged.push(out, tag=it)            # the log line of the previous step (one all-reduce, read one step late)
    if rank == 0 and late is not None:
        print("  logged one step late:", late, flush=True)
    torch.cuda.synchronize()
    if it == 0 and rank == 0:
        red = step._reducer
        print("gradient all-reduce: %d slices started from backward hooks (%s floats), the last one after backward"
              % (len(red.slices), [n for _, n in red.slices]), flush=True)
        assert len(red.slices) == 6 and red._next == 6 and sum(n for _, n in red.slices) == arenas["grad"].numel()
    flat = torch.cat([arenas["param"].detach().view(-1), arenas["target"].detach().view(-1)]).cpu()
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(flat, ref))
    ok = torch.tensor([1.0 if same and bool(torch.isfinite(flat).all()) else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("step %d: parameters and target parameters identical on both ranks and finite: %s" % (it, bool(ok.item())), flush=True)
    if ok.item() != 1.0:
        raise SystemExit(1)
ev[1].record()
torch.cuda.synchronize()
from cstp_amd.train import sync_buffers  # noqa: E402
before = arenas["buffers"].detach().cpu().clone()
sync_buffers(ddp)
buf = arenas["buffers"].detach().cpu()
ref = buf.clone()
dist.broadcast(ref, src=0)
okb = torch.tensor([1.0 if torch.equal(buf, ref) else 0.0])
dist.all_reduce(okb, op=dist.ReduceOp.MIN)
if rank == 0:
    print("BN buffers after sync_buffers identical on both ranks: %s" % bool(okb.item()))
if rank == 1:
    print("rank 1's running statistics differed from rank 0's before the broadcast (per-rank shards): %s" % (not torch.equal(before, buf)))
if okb.item() != 1.0:
    raise SystemExit(1)
if rank == 0:
    print("tile table: %s" % ops.tune_stats)
    print("R(2+1)D-%d, %d ranks x %d clip pairs on ONE MI355X over gloo: %.1f ms per step after the first (not a throughput figure: "
          "both ranks share the GPU and the 177 MB gradient all-reduce goes through host memory)"
          % (depth, world, b_local, ev[0].elapsed_time(ev[1]) / max(steps - 1, 1)))
dist.barrier()
dist.destroy_process_group()
