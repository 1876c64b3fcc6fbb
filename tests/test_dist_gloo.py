"""world_size-2 data-parallel semantics on CPU (gloo): clip sharding, the NT-Xent all-gather with
gradient and its DDP scale factor, and the multi-GPU parity definition of SURVEY 7.2-7 / 8(e)
(per-rank BN statistics; gradient = mean over shards).  Compute here is the CPU oracle -- these
tests cover the host-side distributed logic, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)


def _worker_ntxent(rank, port, q):
    from cstp_amd.ntxent import all_gather_with_grad
    from oracle import r21d_byol_oracle as orc
    _init(rank, port)
    g = torch.Generator().manual_seed(7)
    zi_all = torch.randn(8, 32, generator=g, dtype=torch.float64)
    zj_all = torch.randn(8, 32, generator=g, dtype=torch.float64)
    sl = slice(rank * 4, rank * 4 + 4)
    zi = zi_all[sl].clone().requires_grad_(True)
    zj = zj_all[sl].clone().requires_grad_(True)
    gi, gj = all_gather_with_grad(zi), all_gather_with_grad(zj)
    assert gi.shape == (8, 32) and torch.equal(gi.detach(), zi_all)
    loss = orc.ntxent(gi, gj, 0.5)
    (loss * WORLD).backward()                   # ddp_scale = world_size
    # DDP would now average parameter gradients over ranks; emulate with the embeddings' own grads
    full_i = torch.zeros(8, 32, dtype=torch.float64)
    full_i[sl] = zi.grad
    dist.all_reduce(full_i)
    full_i /= WORLD
    # single-process global-batch reference
    ri = zi_all.clone().requires_grad_(True)
    rj = zj_all.clone().requires_grad_(True)
    ref = orc.ntxent(ri, rj, 0.5)
    ref.backward()
    q.put((rank, float(loss.detach()), float(ref.detach()), float((full_i - ri.grad).abs().max()),
           float((zi.grad / WORLD - ri.grad[sl]).abs().max())))
    dist.destroy_process_group()


def _worker_sharding(rank, port, q):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from cstp_amd.synthetic import SyntheticClips
    _init(rank, port)
    ds = SyntheticClips(length=16, sample_duration=2, sample_size=8, seed=1)
    sampler = DistributedSampler(ds, num_replicas=WORLD, rank=rank, shuffle=True)
    sampler.set_epoch(3)
    global_batch = 8
    loader = DataLoader(ds, batch_size=int(global_batch / WORLD), sampler=sampler, drop_last=True)
    idx = list(iter(sampler))
    batches = [b for b in loader]
    (c1, c2), (spa, tem, pb, (r1, r2)) = batches[0]
    assert c1.shape == (4, 3, 2, 8, 8) and spa.dtype == torch.int64 and r1.shape == (4,)
    # the driver's logging all-reduce (main_byol.py:22-26)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    q.put((rank, idx, len(batches), float(t / WORLD)))
    dist.destroy_process_group()


def _worker_grad_mean(rank, port, q):
    from oracle import r21d_byol_oracle as orc
    _init(rank, port)
    ls = (1, 1, 1, 1)
    x1, x2, labels = orc.closed_form_clips(4, 2, 16, torch.float32)
    w = (0.1, 1.0, 1.0, 1.0, 1.0)

    def shard_grads(r):
        sd = orc.closed_form_state(ls, torch.float32)
        sl = slice(2 * r, 2 * r + 2)
        lab = {k: v[sl] for k, v in labels.items()}
        info = orc.train_step(sd, {}, x1[sl], x2[sl], lab, ls, 0.0, 0.9, 0.0, w, False)
        return info

    mine = shard_grads(rank)
    keys = ["online_net.conv1.spatial_conv.weight", "online_net.conv5.block1.bn2.weight", "predictor.net.3.bias",
            "overlap_spa.3.weight"]
    flat = torch.cat([mine["grads"][k].reshape(-1) for k in keys])
    dist.all_reduce(flat)
    flat /= WORLD                               # what DDP leaves in .grad
    if rank == 0:
        other = shard_grads(1)
        expect = torch.cat([(mine["grads"][k] + other["grads"][k]).reshape(-1) / 2 for k in keys])
        # per-rank BN: a 2+2 split is NOT the 4-clip single-process result
        sd = orc.closed_form_state(ls, torch.float32)
        whole = orc.train_step(sd, {}, x1, x2, labels, ls, 0.0, 0.9, 0.0, w, False)
        q.put((float((flat - expect).abs().max() / expect.abs().max()),
               abs(float(whole["loss_byol"]) - float(mine["loss_byol"]))))
    dist.barrier()
    dist.destroy_process_group()


def _worker_flat_allreduce(rank, port, q):
    from cstp_amd.train import allreduce_mean_
    _init(rank, port)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)       # rank r holds (r+1) * v
    allreduce_mean_(g)
    q.put((rank, float((g - torch.arange(1000, dtype=torch.float32) * 1.5).abs().max())))
    dist.destroy_process_group()


def _run(worker, nres):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(nres)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_ntxent_all_gather_gradient_and_ddp_scale():
    for rank, loss, ref, err_full, err_local in _run(_worker_ntxent, WORLD):
        assert abs(loss - ref) < 1e-12          # every rank evaluates the same global loss
        assert err_full < 1e-12                 # DDP's mean over ranks of grad(world * loss) == global-batch gradient
        assert err_local < 1e-12


def test_flat_gradient_allreduce_is_the_mean_over_ranks():
    for rank, err in _run(_worker_flat_allreduce, WORLD):
        assert err < 1e-6


def test_clip_sharding_and_logging_allreduce():
    res = sorted(_run(_worker_sharding, WORLD))
    idx0, idx1 = res[0][1], res[1][1]
    assert len(idx0) == len(idx1) == 8 and not set(idx0) & set(idx1) and sorted(idx0 + idx1) == list(range(16))
    assert res[0][2] == 2 and res[0][3] == 1.5 and res[1][3] == 1.5


def test_gradient_is_mean_over_shards_with_per_rank_bn():
    (err, loss_gap), = _run(_worker_grad_mean, 1)
    assert err < 1e-6
    assert loss_gap > 1e-6      # BN statistics are per rank: sharded != whole-batch
