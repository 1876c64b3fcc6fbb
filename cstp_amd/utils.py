"""Running meters, the per-epoch TSV log (/root/reference/utils.py:7-48; columns fixed by main_byol.py:216-225 and
main_ft_mp.py:117-131), top-1 accuracy (utils.py:58-66) and the loader construction of utils.py:91-163."""
from __future__ import annotations

import csv

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

LOG_COLUMNS = ["epoch", "loss", "loss_byol", "loss_pred_spa", "loss_pred_tem", "loss_pred_pb", "loss_pred_rot", "acc", "lr"]


class AverageMeter:
    """Tracks the latest value and the sample-weighted running mean."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class Logger:
    """Tab-separated log: header row on overlay=True ('w'), append otherwise."""

    def __init__(self, path, header, overlay=True):
        self.header = list(header)
        self.log_file = open(path, "w" if overlay else "a")
        self.logger = csv.writer(self.log_file, delimiter="\t")
        if overlay:
            self.logger.writerow(self.header)
            self.log_file.flush()

    def log(self, values):
        missing = [c for c in self.header if c not in values]
        if missing:
            raise AssertionError("missing log columns: %s" % missing)
        self.logger.writerow([values[c] for c in self.header])
        self.log_file.flush()

    def close(self):
        self.log_file.close()


def calculate_accuracy(outputs: torch.Tensor, targets: torch.Tensor) -> float:
    """Fraction of rows whose arg-max logit is the target (utils.py:58-66).  One host sync, as in the reference."""
    batch_size = targets.size(0)
    _, pred = outputs.topk(1, 1, True)
    correct = pred.t().eq(targets.view(1, -1))
    return correct.float().sum().item() / batch_size


def get_dataloader(dataset, opts, data_type="train"):
    """utils.py:91-163: the GLOBAL --batch_size is split over ranks; 'byol'/'train' shuffle and drop the last partial
    batch, 'val' keeps order and keeps it.  Returns (loader, sampler); sampler is None without DDP."""
    if data_type not in ("byol", "train", "val"):
        raise ValueError("data_type %r" % (data_type,))
    train = data_type in ("byol", "train")
    if getattr(opts, "distributed", False):
        sampler = DistributedSampler(dataset, num_replicas=opts.world_size, rank=opts.rank, shuffle=train)
        batch_size = int(opts.batch_size / opts.world_size)
        if data_type == "byol":
            opts.batch_size = batch_size     # the reference overwrites the option on this branch only (utils.py:98)
        shuffle = False
    else:
        sampler, batch_size, shuffle = None, opts.batch_size, train
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=opts.n_workers, pin_memory=True,
                        sampler=sampler, drop_last=train)
    return loader, sampler
