"""Command-line surface of the pre-training driver: the same flat flag namespace as
/root/reference/opts.py:4-245 (same flag names, defaults and types, so the reference's launch
scripts -- README.md:41-50, script/r2p1d/kin400/*.sh -- run unchanged), expressed as a table.

Additions (all optional, defaults reproduce the reference):
  --ntxent_weight   weight of the NT-Xent term on all-gathered projector embeddings (default 0:
                    the reference builds the criterion but never adds it, main_byol.py:71-73,191-197)
  --synthetic_len   samples per epoch of the built-in synthetic dataset (--dataset synthetic)
  --max_steps       stop an epoch after this many iterations (0 = full epoch)
  --bucket_cap_mb   DDP gradient bucket size (xGMI ring all-reduce is per-link bound)
  --act_dtype       fp32 (default) | bf16: storage type of the 5-D activations and their gradients (BASELINE configs[4]:
                    3D-ResNet-50, bf16; --model_name r3d_byol -- cstp_amd/r3d_byol.py)
torchrun passes LOCAL_RANK through the environment instead of --local_rank; both are honoured.
"""
from __future__ import annotations

import argparse
import os

# (flag, default, type, help) -- type None means store_true
_FLAGS = [
    # datasets
    ("frame_dir", "dataset/HMDB51/", str, "path of jpg files"),
    ("annotation_path", "dataset/HMDB51_labels", str, "label paths"),
    ("dataset", "HMDB51", str, "HMDB51 | UCF101 | Kinetics | synthetic | synthetic_video"),
    ("split", 1, str, "split id (HMDB51 / UCF101)"),
    ("modality", "RGB", str, "RGB | Flow"),
    ("input_channels", 3, int, "3 | 2"),
    ("n_classes", 400, int, "number of classes"),
    ("n_finetune_classes", 51, int, "number of classes when fine-tuning"),
    # model
    ("model_name", "resnext", str, "backbone (this package: r21d_byol)"),
    ("model_depth", 101, int, "r21d_byol: 1 -> (1,1,1,1), 18 -> (2,2,2,2), 34 -> (3,4,6,3)"),
    ("resnet_shortcut", "B", str, "shortcut type of resnet (A | B)"),
    ("resnext_cardinality", 32, int, "ResNeXt cardinality"),
    ("ft_begin_index", 0, int, "first block to fine-tune"),
    ("sample_size", 112, int, "clip height and width"),
    ("sample_duration", 16, int, "clip length in frames"),
    ("batch_size", 32, int, "GLOBAL batch size (split over ranks)"),
    ("n_workers", 4, int, "dataloader workers"),
    ("pretrained_path", "", str, "pretrained checkpoint"),
    ("test_md_path", "", str, "checkpoint to test"),
    ("resume_md_path", "", str, "checkpoint to resume"),
    # optimiser
    ("learning_rate", 3e-4, float, "peak learning rate"),
    ("momentum", 0.9, float, "SGD momentum"),
    ("dampening", 0.9, float, "accepted for compatibility; the reference never passes it to SGD"),
    ("weight_decay", 1e-4, float, "weight decay"),
    ("nesterov", False, None, "accepted for compatibility; unused by the reference driver"),
    ("optimizer", "sgd", str, "sgd | adamw | adam (flat-arena HIP kernels)"),
    ("lr_patience", 10, int, "ReduceLROnPlateau patience (fine-tune only)"),
    ("n_epochs", 400, int, "epochs"),
    # logging / misc
    ("result_path", "", str, "output directory"),
    ("log", True, None, "kept for compatibility"),
    ("manual_seed", 1, int, "random seed"),
    ("random_seed", 1, bool, "kept for compatibility"),
    ("cuda", False, None, "set by the driver"),
    ("device", None, str, "set by the driver"),
    ("tau", 8, int, "slow-path stride (other backbones)"),
    ("alpha", 4, int, "fast/slow frame-rate ratio (other backbones)"),
    ("input_h", 128, int, "input height before crop"),
    ("input_w", 171, int, "input width before crop"),
    ("temperature", 0.5, float, "NT-Xent temperature"),
    ("task", "r_ctr", str, "loss_com | r_byol | ..."),
    ("temp_transform", "speed/random/periodic/warp", str, "temporal transforms"),
    ("lr_decay", 1e-4, float, "learning rate decay"),
    ("local_rank", -1, int, "GPU rank (torch.distributed.launch); env LOCAL_RANK also honoured"),
    ("rank", -1, int, "process rank"),
    ("dist_url", "env://", str, "rendezvous url"),
    ("dist_backend", "nccl", str, "nccl (= RCCL on ROCm) | gloo"),
    ("world_size", -1, int, "set by the driver from WORLD_SIZE"),
    ("nprocs", -1, int, "set by the driver"),
    ("distributed", False, None, "set by the driver"),
    ("sync_bn", 1, int, "kept: the reference's SyncBN spans a one-rank group, i.e. per-GPU BN either way"),
    ("clip_grad_norm", 1, int, "1 = clip_grad_norm_(., 18)"),
    ("split_path", "", str, "training list path"),
    ("pb_rate", 4, int, "playback rate of a clip 1,2,4,8"),
    ("transform_mode", "numpy", str, "transform mode"),
    ("input_size", 320, int, "input size"),
    ("output_feat", 128, int, "output feature size"),
    ("norm_method", "tf_norm", str, "input normalisation"),
    ("max_iter", 80000, int, "maximum iterations"),
    ("t_ft_task", "", str, "fine-tune task for test"),
    ("sc_type", "B", str, "resnet shortcut type"),
    ("lmdb_path", "", str, "LMDB path"),
    # additions
    ("ntxent_weight", 0.0, float, "weight of the NT-Xent term on all-gathered embeddings (0 = reference)"),
    ("synthetic_len", 256, int, "samples per epoch for --dataset synthetic"),
    ("max_steps", 0, int, "stop each epoch after this many iterations (0 = all)"),
    ("bucket_cap_mb", 25, int, "DDP gradient bucket size in MB"),
    ("act_dtype", "fp32", str, "fp32 | bf16: activation storage type (bf16: --model_name r3d_byol)"),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="CSTP pre-training on MI355X")
    for name, default, typ, text in _FLAGS:
        if typ is None:
            parser.add_argument("--" + name, action="store_true", help=text)
            parser.set_defaults(**{name: default})
        else:
            parser.add_argument("--" + name, default=default, type=typ, help=text)
    parser.add_argument("--highest_val", default={"name": 0}, type=dict, help="best validation score store")
    parser.add_argument("--loss_weight", default=1.0, nargs="+", type=float,
                        help="weights of (byol, spatial overlap, temporal overlap, playback rate, rotation)")
    return parser


def parse_opts(argv=None):
    args = build_parser().parse_args(argv)
    if args.local_rank == -1 and "LOCAL_RANK" in os.environ:   # torchrun / torch.distributed.run
        args.local_rank = int(os.environ["LOCAL_RANK"])
    return args
