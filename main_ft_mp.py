#!/usr/bin/env python3
"""CSTP fine-tune / train-from-scratch driver for MI355X -- drop-in for the reference's main_ft_mp.py.

    python -m torch.distributed.run --nproc_per_node=8 --master-addr 127.0.0.1 main_ft_mp.py \
        --dataset synthetic --n_classes 101 --batch_size 64 --sample_duration 16 --sample_size 112 \
        --model_name r21d_byol --model_depth 18 --task ft_all --pretrained_path results/synthetic/loss_com/save_300.pth \
        --learning_rate 0.01 --weight_decay 5e-4 --n_epochs 30 --result_path results

What is kept from /root/reference/main_ft_mp.py: seeding (:29-31), env:// process group and rank-0-only printing
(:51-62), train/val loaders with the global batch split over ranks (:75-102, utils.py:91-163), generate_model incl. the
checkpoint handling per task (:106), sgd / adamw(betas 0.9,0.99) / adam (:133-147), ReduceLROnPlateau('min',
patience=--lr_patience) stepped with the epoch's validation loss (:153,279), the train loop (:178-242: forward with
o_type=task, CrossEntropy, accuracy, loss all-reduce for the meter, zero_grad/backward/step, the print columns, the TSV row),
the validation loop under model.eval() + no_grad (:245-310) with the best-accuracy checkpoint ``save_{epoch}_max.pth``
replacing the previous best, and the side-stream host-to-HBM batch overlap (:313-352; cstp_amd.device_batches).
What differs (each a fix of something that cannot work in the reference, none changes the arithmetic):
  * ``--task scratch`` forwards with o_type 'ft_all' (the reference passes o_type='scratch', which R21DBYOL.forward
    rejects, r21d_byol.py:400-401); ``--task resume`` continues an ft_all run (undefined model in the reference);
  * the validation loss is averaged over ranks and EVERY rank steps the plateau scheduler (the reference steps it on
    rank 0 only, :278-279, so after the first reduction the ranks train with different learning rates);
  * --dataset synthetic (class-patterned clips) stands in for the out-of-scope UCF/Kinetics readers.
"""
from __future__ import annotations

import builtins
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from cstp_amd import ops
from cstp_amd.model import generate_model
from cstp_amd.optim import build_optimizer
from cstp_amd.opts import parse_opts
from cstp_amd.device_batches import DeviceBatches
from cstp_amd.scheduler import ReduceLROnPlateau
from cstp_amd.synthetic import SyntheticLabelledClips
from cstp_amd.train import FineTuneStep, sync_buffers
from cstp_amd.utils import AverageMeter, Logger, calculate_accuracy, get_dataloader

TRAIN_TASKS = ("ft_fc", "ft_all", "scratch", "resume")


def reduce_mean(tensor, world_size):
    rt = tensor.clone()
    if dist.is_initialized():
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= world_size
    return rt


def build_dataset(opts, data_type):
    if opts.dataset != "synthetic":
        raise NotImplementedError("dataset %r: only --dataset synthetic is built in (the reference's UCF/Kinetics readers "
                                  "are outside this package's scope)" % opts.dataset)
    length = opts.synthetic_len if data_type == "train" else max(opts.synthetic_len // 4, 1)
    return SyntheticLabelledClips(data_type, length, opts.sample_duration, opts.sample_size, opts.n_classes,
                                  opts.manual_seed)


def o_type_for(task):
    return "ft_all" if task in ("scratch", "resume") else task


def train(epoch, train_dataloader, step_fn, optimizer, opts, train_logger, len_train_data):
    step_fn.model.train()
    batch_time, data_time, losses, accuracies = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    end_time = time.time()
    n_iter = int(len_train_data / opts.batch_size)
    i = 0
    for inputs, targets in DeviceBatches(train_dataloader, opts.device):
        i += 1
        data_time.update(time.time() - end_time)
        loss, outputs = step_fn(inputs, targets)
        acc = calculate_accuracy(outputs, targets)
        reduced_loss = reduce_mean(loss, opts.world_size)
        losses.update(reduced_loss.item(), inputs.size(0))
        accuracies.update(acc, inputs.size(0))
        batch_time.update(time.time() - end_time)
        end_time = time.time()
        print("Epoch: [{0}][{1}/{2}]\t"
              "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
              "Data {data_time.val:.3f} ({data_time.avg:.3f})\t"
              "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
              "Acc {acc.val:.3f} ({acc.avg:.3f})\t"
              "Lr {lr:.6f}\t"
              "Left {left:.1f}d".format(epoch, i, n_iter, batch_time=batch_time, data_time=data_time, loss=losses,
                                        acc=accuracies, lr=optimizer.param_groups[-1]["lr"],
                                        left=(batch_time.avg * ((opts.n_epochs - epoch) * n_iter + n_iter - i)) / 3600 / 24))
        if opts.max_steps and i >= opts.max_steps:
            break
    if opts.rank == 0 and opts.local_rank == 0:
        train_logger.log({"epoch": epoch, "loss": losses.avg, "acc": accuracies.avg,
                          "lr": float("{:.5f}".format(optimizer.param_groups[-1]["lr"]))})
    return losses.avg, accuracies.avg


def validation(epoch, val_dataloader, model, optimizer, opts, val_logger, len_val_data, scheduler):
    batch_time, data_time, losses, accuracies = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    o_type = o_type_for(opts.task)
    n_iter = int(len_val_data / opts.batch_size)
    model.eval()     # BatchNorm switches to its running statistics (cstp_bn_forward_eval)
    # DDP(broadcast_buffers=True) hands rank 0's running statistics to every rank at the first eval forward too
    # (models/model.py:97-103): every rank validates -- and rank 0 checkpoints -- the same statistics
    sync_buffers(model)
    with torch.no_grad():
        i = 0
        for inputs, targets in DeviceBatches(val_dataloader, opts.device):
            end_time = time.time()
            outputs = model(inputs, o_type=o_type)
            loss = ops.cross_entropy(outputs, targets)
            acc = calculate_accuracy(outputs, targets)
            losses.update(loss.item(), inputs.size(0))
            accuracies.update(acc, inputs.size(0))
            batch_time.update(time.time() - end_time)
            print("Val_Epoch: [{0}][{1}/{2}]\t"
                  "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                  "Data {data_time.val:.3f} ({data_time.avg:.3f})\t"
                  "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                  "Acc {acc.val:.3f} ({acc.avg:.3f})".format(epoch, i + 1, n_iter, batch_time=batch_time,
                                                             data_time=data_time, loss=losses, acc=accuracies))
            i += 1
    # every rank sees the same plateau metric and takes the same lr decision (see the module docstring)
    val_loss = losses.avg
    if opts.distributed:
        t = torch.tensor([losses.sum, float(losses.count)], dtype=torch.float64, device=torch.device("cuda", opts.device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        val_loss = float(t[0] / t[1].clamp(min=1))
    scheduler.step(val_loss)
    if opts.rank == 0 and opts.local_rank == 0:
        val_logger.log({"epoch": epoch, "loss": losses.avg, "acc": accuracies.avg})
        accuracy_val = accuracies.avg
        if accuracy_val > list(opts.highest_val.values())[0]:
            old_key = list(opts.highest_val.keys())[0]
            file_path = os.path.join(opts.result_path, opts.dataset, opts.task, old_key)
            if os.path.exists(file_path):
                os.remove(file_path)
            opts.highest_val.pop(old_key)
            opts.highest_val["save_{}_max.pth".format(epoch)] = accuracy_val
            save_file_path = os.path.join(opts.result_path, opts.dataset, opts.task, "save_{}_max.pth".format(epoch))
            torch.save({"epoch": epoch + 1, "arch": opts.arch, "state_dict": model.state_dict(),
                        "optimizer": optimizer.state_dict()}, save_file_path)
    return val_loss, accuracies.avg


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
    if opts.task not in TRAIN_TASKS:
        raise ValueError("main_ft_mp.py serves --task %s, got %r" % ("/".join(TRAIN_TASKS), opts.task))

    print("Preprocessing train data ...")
    train_data = build_dataset(opts, "train")
    len_train_data = len(train_data)
    print("Length of training data = ", len_train_data)
    train_dataloader, train_sampler = get_dataloader(train_data, opts=opts, data_type="train")
    print("Preprocessing validation data ...")
    val_data = build_dataset(opts, "val")
    len_val_data = len(val_data)
    print("Length of validation data = ", len_val_data)
    val_dataloader, _ = get_dataloader(val_data, opts=opts, data_type="val")

    print("Loading model... ", opts.model_name, opts.model_depth)
    model, parameters = generate_model(opts)
    inner = model.module

    resume = opts.task == "resume"
    begin_epoch = int(opts.resume_md_path.split("/")[-1].split("_")[1]) if resume else 1
    name = "{}_{}_clip{}model{}{}.log"
    train_logger = val_logger = None
    if local_rank == 0:
        train_logger = Logger(os.path.join(log_path, name.format(opts.dataset, "train", opts.sample_duration, opts.model_name,
                                                                 opts.model_depth)),
                              ["epoch", "loss", "acc", "lr"], overlay=not resume)
        val_logger = Logger(os.path.join(log_path, name.format(opts.dataset, "val", opts.sample_duration, opts.model_name,
                                                               opts.model_depth)),
                            ["epoch", "loss", "acc"], overlay=not resume)

    optimizer = build_optimizer(opts, parameters, inner.flatten_parameters())
    if resume:
        optimizer.load_state_dict(torch.load(opts.resume_md_path, map_location=torch.device("cuda", local_rank))["optimizer"])
    scheduler = ReduceLROnPlateau(optimizer, "min", patience=opts.lr_patience)
    step_fn = FineTuneStep(model, optimizer, o_type_for(opts.task))

    for epoch in range(begin_epoch, opts.n_epochs + 1):
        print("Start to fine-tune")
        print("Start training epoch {}".format(epoch))
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        train(epoch, train_dataloader, step_fn, optimizer, opts, train_logger, len_train_data)
        print("Start validating epoch {}".format(epoch))
        validation(epoch, val_dataloader, model, optimizer, opts, val_logger, len_val_data, scheduler)
    if opts.distributed:
        dist.barrier()
        dist.destroy_process_group()


def main(opts):
    torch.manual_seed(opts.manual_seed)
    np.random.seed(opts.manual_seed)
    random.seed(opts.manual_seed)
    if not torch.cuda.is_available():
        raise RuntimeError("main_ft_mp.py needs a HIP device: cstp_amd has no CPU execution path")
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
