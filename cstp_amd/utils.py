"""Running meters and the per-epoch TSV log of the pre-training driver
(/root/reference/utils.py:7-48; columns fixed by main_byol.py:216-225)."""
from __future__ import annotations

import csv

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
