#!/usr/bin/env python3
"""CSTP BYOL pre-training driver for MI355X -- drop-in for the reference's main_byol.py.

Launch exactly like the reference (README.md:41-50), one process per GPU over RCCL:

    python -m torch.distributed.run --nproc_per_node=8 --master-addr 127.0.0.1 main_byol.py \
        --dataset synthetic --batch_size 128 --sample_duration 16 --model_name r21d_byol \
        --model_depth 18 --n_epochs 300 --learning_rate 0.09 --weight_decay 5e-4 --sample_size 112 \
        --task loss_com --optimizer sgd --loss_weight 0.1 1 1 1 1 --result_path results

What is kept from /root/reference/main_byol.py: seeding (:144-146), env:// process group on
--dist_backend (:171-174), rank-0-only printing (:166-169), global --batch_size split over ranks
(utils.py:111), generate_model -> DDP (:211), SGD(momentum, wd) (:228-232), the per-epoch
cosine/warm-up schedule starting at 1e-5 (:252-258,269), the per-step loss composition, clip at 18
and optimiser step (:60-91), the per-iteration print columns (:96-117), the per-epoch TSV row
(:119-130) and the save_{epoch}.pth checkpoint dict every 100 epochs (:132-140).
What differs: --dataset synthetic feeds random clips through a DataLoader; --dataset synthetic_video keeps decoded
uint8 videos in HBM and samples / rotates / crops / resizes / flips / normalises the clip pairs on the GPU
(cstp_amd.sampler + cstp_clip_assemble) in place of the reference's PIL worker pipeline; reading JPEG / LMDB data
sets is out of scope; the log scalars reach the host through one pinned-memory copy per iteration, read one step late
(cstp_amd.train.LaggedScalars), instead of seven .item() syncs and a blocking all-reduce per iteration.
"""
from __future__ import annotations

import builtins
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from cstp_amd.model import generate_model
from cstp_amd.ntxent import NTXentLoss
from cstp_amd.optim import build_optimizer
from cstp_amd.opts import parse_opts
from cstp_amd.scheduler import CosineAnnealingWarmupRestarts
from cstp_amd.synthetic import SyntheticClips
from cstp_amd.train import LaggedScalars, PretrainStep
from cstp_amd.utils import LOG_COLUMNS, AverageMeter, Logger


def build_dataset(opts):
    if opts.dataset == "synthetic_video":
        from cstp_amd.clip_ops import GpuVideoClips
        return GpuVideoClips(torch.device("cuda", opts.local_rank), sample_duration=opts.sample_duration,
                             sample_size=opts.sample_size, length=opts.synthetic_len, seed=opts.manual_seed)
    if opts.dataset != "synthetic":
        raise NotImplementedError("dataset %r: --dataset synthetic and synthetic_video are built in (reading the reference's "
                                  "JPEG / LMDB data sets is outside this package's scope)" % opts.dataset)
    return SyntheticClips(opts.synthetic_len, opts.sample_duration, opts.sample_size, opts.manual_seed)


def train_BYOL(epoch, loader, step_fn, optimizer, opts, train_logger):
    meters = {k: AverageMeter() for k in ("batch", "data", "loss", "loss_byol", "loss_pred_spa", "loss_pred_tem",
                                          "loss_pred_pb", "loss_pred_rot")}
    dev = torch.device("cuda", opts.local_rank)
    # The log scalars (and the rank-mean of the total loss, main_byol.py:22-26,75) lag ONE step behind the launches: no
    # host sync and no blocking collective between two steps; the columns are the reference's.
    lagged = LaggedScalars(dev, opts.world_size)

    def log_line(rec):
        (it, n, t_batch, t_data), host = rec
        for k in LaggedScalars.KEYS:
            meters[k].update(host[k], n)
        meters["batch"].update(t_batch)
        meters["data"].update(t_data)
        m = meters
        print("Epoch: [{0}][{1}/{2}]\tTime {3:.3f} ({4:.3f})\tData {5:.3f} ({6:.3f})\t"
              "Loss_byol {7:.4f} ({8:.4f})\tLoss_pred_spa {9:.4f} ({10:.4f})\tLoss_pred_tem {11:.4f} ({12:.4f})\t"
              "Loss_pred_pb {13:.4f} ({14:.4f})\tLoss_pred_rot {15:4f} ({16:.4f})Loss_total {17:.4f} ({18:.4f})\t"
              "Lr {19:.4}".format(epoch, it, len(loader), m["batch"].val, m["batch"].avg, m["data"].val, m["data"].avg,
                                  m["loss_byol"].val, m["loss_byol"].avg, m["loss_pred_spa"].val, m["loss_pred_spa"].avg,
                                  m["loss_pred_tem"].val, m["loss_pred_tem"].avg, m["loss_pred_pb"].val,
                                  m["loss_pred_pb"].avg, m["loss_pred_rot"].val, m["loss_pred_rot"].avg, m["loss"].val,
                                  m["loss"].avg, optimizer.param_groups[-1]["lr"]))

    end = time.time()
    for i, (inputs, targets) in enumerate(loader):
        if opts.max_steps and i >= opts.max_steps:
            break
        t_data = time.time() - end
        clip_1 = inputs[0].to(dev, non_blocking=True)
        clip_2 = inputs[1].to(dev, non_blocking=True)
        spa, tem, pb = (targets[j].to(dev, non_blocking=True) for j in range(3))
        rot_1, rot_2 = targets[3][0].to(dev, non_blocking=True), targets[3][1].to(dev, non_blocking=True)
        out = step_fn(clip_1, clip_2, spa, tem, pb, rot_1, rot_2)
        t_batch = time.time() - end
        end = time.time()
        prev = lagged.push(out, (i + 1, clip_1.size(0), t_batch, t_data))
        if prev is not None:
            log_line(prev)
    last = lagged.flush()
    if last is not None:
        log_line(last)
    if opts.local_rank == 0:
        row = {k: meters[k].avg for k in LOG_COLUMNS if k in meters}
        row.update({"epoch": epoch, "acc": None, "lr": float("{:.5f}".format(optimizer.param_groups[-1]["lr"]))})
        train_logger.log(row)
        if opts.rank == 0 and epoch % 100 == 0:     # rank 0 writes its own state: what DDP's broadcast hands every rank
            path = os.path.join(opts.result_path, opts.dataset, opts.task, "save_{}.pth".format(epoch))
            torch.save({"epoch": epoch + 1, "arch": opts.arch, "state_dict": step_fn.model.state_dict(),
                        "optimizer": optimizer.state_dict()}, path)


def main_worker(local_rank, opts):
    opts.device = local_rank
    if opts.distributed:
        if local_rank != 0:
            builtins.print = lambda *a, **k: None   # only the master prints
        opts.rank = local_rank                      # single-node assumption, as in the reference
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=opts.dist_backend, init_method=opts.dist_url, world_size=opts.world_size,
                                rank=opts.rank)
    log_path = os.path.join(opts.result_path, opts.dataset, opts.task)
    if local_rank == 0:
        os.makedirs(log_path, exist_ok=True)
    print(opts)
    opts.arch = "{}-{}".format(opts.model_name, opts.model_depth)

    criterion_ctr = NTXentLoss(device=local_rank, batch_size=opts.batch_size, temperature=opts.temperature,
                               use_cosine_similarity=True)
    train_data = build_dataset(opts)
    print("Length of training data = ", len(train_data))
    per_rank = int(opts.batch_size / opts.world_size)
    if opts.dataset == "synthetic_video":
        from cstp_amd.clip_ops import GpuClipLoader
        loader = sampler = GpuClipLoader(train_data, per_rank, rank=max(opts.rank, 0), world_size=opts.world_size,
                                         seed=opts.manual_seed)
    else:
        sampler = DistributedSampler(train_data, num_replicas=opts.world_size, rank=max(opts.rank, 0), shuffle=True) \
            if opts.distributed else None
        loader = DataLoader(train_data, batch_size=per_rank, shuffle=sampler is None, num_workers=opts.n_workers,
                            pin_memory=True, sampler=sampler, drop_last=True)

    print("Loading model... ", opts.model_name, opts.model_depth)
    model, parameters = generate_model(opts)
    print("Model is loaded successfully!")
    inner = model.module if hasattr(model, "module") else model
    train_logger = Logger(os.path.join(log_path, "{}_train_clip{}model{}{}.log".format(
        opts.dataset, opts.sample_duration, opts.model_name, opts.model_depth)), LOG_COLUMNS,
        overlay=not opts.resume_md_path) \
        if local_rank == 0 else None

    optimizer = build_optimizer(opts, parameters, inner.flatten_parameters())   # sgd | adamw | adam (main_byol.py:227-244)
    begin_epoch = 1
    if opts.resume_md_path:
        # The reference only reloads the optimizer for --task resume and then trains nothing (main_byol.py:246-247,
        # 260).  Here --resume_md_path continues a loss_com run: weights (online, target, heads, BN buffers), momentum
        # buffers, the epoch counter (checkpoints store epoch + 1) and the schedule position.
        md = torch.load(opts.resume_md_path, map_location=torch.device("cuda", local_rank))
        assert opts.arch == md["arch"]
        model.load_state_dict(md["state_dict"])
        optimizer.load_state_dict(md["optimizer"])
        begin_epoch = int(md["epoch"])
        print("Resume model {} at epoch {}".format(opts.resume_md_path, begin_epoch))
    scheduler = CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=opts.n_epochs, cycle_mult=1.0,
                                              max_lr=opts.learning_rate, min_lr=0.00001,
                                              warmup_steps=0.5 * opts.n_epochs, gamma=0.5)
    for _ in range(1, begin_epoch):
        scheduler.step()
    step_fn = PretrainStep(model, optimizer, opts.loss_weight, task=opts.task, clip_grad_norm=opts.clip_grad_norm,
                           ntxent=criterion_ctr, ntxent_weight=opts.ntxent_weight)
    if opts.task in ("r_byol", "loss_com"):
        print("Start to train BYOL CoCLR data augmentation pre-trained model!")
        for epoch in range(begin_epoch, opts.n_epochs + 1):
            print("Training BYOL at epoch {}".format(epoch))
            if sampler is not None:
                sampler.set_epoch(epoch)
            model.train()
            train_BYOL(epoch, loader, step_fn, optimizer, opts, train_logger)
            scheduler.step()
    if opts.distributed:
        dist.barrier()
        dist.destroy_process_group()


def main(opts):
    torch.manual_seed(opts.manual_seed)
    np.random.seed(opts.manual_seed)
    random.seed(opts.manual_seed)
    if not torch.cuda.is_available():
        raise NotImplementedError("Only DistributedDataParallel on HIP devices is supported.")
    opts.cuda = True
    if opts.local_rank != -1:
        opts.world_size = int(os.environ["WORLD_SIZE"])
        opts.distributed = True
        opts.nprocs = torch.cuda.device_count()
        main_worker(opts.local_rank, opts)
    else:
        opts.distributed = False
        opts.world_size = 1
        opts.local_rank = 0
        opts.rank = 0
        main_worker(0, opts)


if __name__ == "__main__":
    main(parse_opts())
