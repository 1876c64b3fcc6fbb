#!/usr/bin/env python3
"""Video-level test of a fine-tuned checkpoint on MI355X -- drop-in for the reference's test.py:23-98.

    python test.py --dataset synthetic --n_classes 101 --model_name r21d_byol --model_depth 18 --task test \
        --t_ft_task ft_all --result_path results --sample_duration 16 --sample_size 112

Kept: single process / single device, DataLoader(batch_size, shuffle=False) over videos whose item is
[n_clips, 3, T, H, W] (squeezed from the batch-of-one), ``opts.test_md_path`` defaulting to the one ``*_max.pth`` under
result_path/dataset/t_ft_task (:51-56), generate_model(task 'test') -> strict checkpoint load, model.eval() + no_grad,
clip logits averaged per video, top-5 from the mean, running top-1 accuracy, one line per video and the final
"Video accuracy" written to test_{model}{depth}_{dataset}_{split}_{modality}_{T}_plusone.txt (:65-98).
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from cstp_amd.model import generate_model
from cstp_amd.opts import parse_opts
from cstp_amd.synthetic import SyntheticLabelledClips
from cstp_amd.utils import AverageMeter


def build_dataset(opts):
    if opts.dataset != "synthetic":
        raise NotImplementedError("dataset %r: only --dataset synthetic is built in" % opts.dataset)
    return SyntheticLabelledClips("test", max(opts.synthetic_len // 4, 1), opts.sample_duration, opts.sample_size,
                                  opts.n_classes, opts.manual_seed)


def video_prediction(model, inputs, o_type="test"):
    """Mean of the per-clip logits of one video and its top-5 class ids (test.py:81-82)."""
    outputs = model(inputs, None, o_type=o_type)
    mean = torch.mean(outputs, dim=0, keepdim=True)
    return mean, np.array(mean.topk(min(5, mean.shape[1]), 1, True)[1].cpu().data[0])


def run(opts):
    if not torch.cuda.is_available():
        raise RuntimeError("test.py needs a HIP device: cstp_amd has no CPU execution path")
    opts.cuda = True
    opts.distributed = False
    opts.local_rank = 0
    opts.device = torch.device("cuda:0")
    print(opts)
    opts.arch = "{}-{}".format(opts.model_name, opts.model_depth)
    print("Preprocessing testing data ...")
    test_data = build_dataset(opts)
    print("Length of testing data = ", len(test_data))
    test_dataloader = DataLoader(test_data, batch_size=1, shuffle=False, num_workers=opts.n_workers, pin_memory=True,
                                 drop_last=False)
    print("Length of test datatloader = ", len(test_dataloader))
    if not opts.test_md_path:
        found = glob.glob(os.path.join(opts.result_path, opts.dataset, opts.t_ft_task, "*_max.pth"))
        if len(found) > 1:
            raise ValueError("Too many models in result path")
        opts.test_md_path = found[0]
    model = generate_model(opts)
    accuracies = AverageMeter()
    result_path = "{}/{}/".format(opts.result_path, opts.dataset)
    os.makedirs(result_path, exist_ok=True)
    out_name = "test_{}{}_{}_{}_{}_{}_plusone.txt".format(opts.model_name, opts.model_depth, opts.dataset, opts.split,
                                                         opts.modality, opts.sample_duration)
    with open(os.path.join(result_path, out_name), "w+") as f:
        f.write(str(opts) + "\n")
        model.eval()
        with torch.no_grad():
            for i, (inputs, labels) in enumerate(test_dataloader):
                inputs = torch.squeeze(inputs, 0).to(opts.device, non_blocking=True)
                labels = labels.to(opts.device, non_blocking=True)
                _, pred5 = video_prediction(model, inputs, opts.task)
                acc = float(pred5[0] == int(labels[0]))
                accuracies.update(acc, 1)
                line = "Video[{}]:\ttop5 = {}\ttop1 = {}\tgt = {}\tacc = {}".format(i, pred5, pred5[0], int(labels[0]),
                                                                                 accuracies.avg)
                print(line)
                f.write(line + "\n")
                f.flush()
        print("Video accuracy = ", accuracies.avg)
        f.write("Video accuracy = " + str(accuracies.avg) + "\n")
    return accuracies.avg


if __name__ == "__main__":
    run(parse_opts())
