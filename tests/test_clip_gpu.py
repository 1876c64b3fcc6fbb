"""cstp_clip_assemble (HIP) against the data-path oracle (oracle/pil_ops.py, pinned to Pillow in test_clip_oracle.py) and against
Pillow itself: the fp32 clip tensors must be IDENTICAL -- the resize is 22-bit integer arithmetic, and the normalisation is the
same three fp32 operations."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import pil_ops as po

pytestmark = pytest.mark.gpu

PIL_ROT = {90: Image.ROTATE_90, 180: Image.ROTATE_180, 270: Image.ROTATE_270}


def _frames(f, h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
    base[:, : h // 2, : w // 2] //= 4            # some structure: dark quadrant, saturated stripe
    base[:, :, w // 3] = 255
    return base


@pytest.mark.parametrize("rot", [0, 90, 180, 270])
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("h,w,box,size", [
    (128, 171, (10, 20, 90, 100), 112),      # upscale 80x80 -> 112
    (240, 320, (0, 0, 240, 240), 112),       # downscale, wide filter support
    (60, 80, (7, 3, 40, 57), 32),            # mixed: up in x, down in y
    (64, 64, (0, 0, 64, 64), 64),            # identity-sized
    (50, 70, (30, 20, 75, 68), 40),          # box reaching past the right / bottom edge in every orientation: zero fill
])
def test_clip_assemble_is_bit_exact(rot, flip, h, w, box, size):
    from cstp_amd import clip_ops, sampler
    frames = _frames(5, h, w, seed=h + w + rot)
    idx = [4, 0, 2, 2]
    rw, rh = (h, w) if rot in (90, 270) else (w, h)
    if box[0] >= rw or box[1] >= rh:
        pytest.skip("box entirely outside this orientation")
    plan = sampler.ClipPlan(idx, rot, box, flip, False)
    got = clip_ops.assemble_clip(torch.from_numpy(frames).cuda(), plan, size).cpu().numpy()
    want = po.assemble_clip(frames, idx, rot, box, size, flip)
    assert got.shape == want.shape == (3, 4, size, size) and got.dtype == np.float32
    assert np.array_equal(got, want)
    # and against Pillow directly, chained as datasets.py:929-948 + preprocess_data.py:513-514,578-581,358-364 chain it
    ref = []
    for f in idx:
        im = Image.fromarray(frames[f], "RGB")
        if rot:
            im = im.transpose(PIL_ROT[rot])
        im = im.crop(box).resize((size, size), Image.BICUBIC)
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)      # ToTensor
        ref.append(torch.clamp(t * 2 - 1, -1, 1).numpy())
    assert np.array_equal(got, np.stack(ref, axis=1))


def test_sampled_pairs_match_the_oracle_and_feed_the_model():
    from cstp_amd import clip_ops, sampler
    frames = _frames(70, 96, 128, seed=3)
    dev = torch.from_numpy(frames).cuda()
    for seed in range(12):
        plan = sampler.sample_pair(70, 128, 96, 16, random.Random(seed))
        (c1, c2), labels = clip_ops.assemble_pair(dev, plan, 64)
        for c, p in ((c1, plan.clip_1), (c2, plan.clip_2)):
            assert np.array_equal(c.cpu().numpy(), po.assemble_clip(frames, p.frames, p.rotate, p.box, 64, p.flip))
        assert labels == [plan.spa_label, plan.tem_label, plan.pb_label, list(plan.rot_labels)]


def test_gpu_video_dataset_batches():
    from cstp_amd import clip_ops
    ds = clip_ops.GpuVideoClips("cuda:0", n_videos=2, frames=40, height=64, width=86, sample_duration=8, sample_size=32, length=16)
    c1, c2, spa, tem, pb, r1, r2 = ds.batch([0, 1, 2, 3])
    assert c1.shape == c2.shape == (4, 3, 8, 32, 32) and c1.dtype == torch.float32 and c1.is_cuda
    assert float(c1.min()) >= -1 and float(c1.max()) <= 1 and float(c1.std()) > 0.05
    for lab, hi in ((spa, 4), (tem, 4), (pb, 3), (r1, 3), (r2, 3)):
        assert lab.dtype == torch.int64 and int(lab.min()) >= 0 and int(lab.max()) <= hi
    again = ds.batch([0, 1, 2, 3])
    assert torch.equal(again[0], c1) and torch.equal(again[3], tem)      # a sample is a function of (seed, index)


def test_clip_assemble_refuses_host_tensors():
    from cstp_amd import clip_ops, sampler, _lib
    with pytest.raises(_lib.CstpError):
        clip_ops.assemble_clip(torch.zeros((2, 8, 8, 3), dtype=torch.uint8), sampler.ClipPlan([0], 0, (0, 0, 8, 8), False, False), 4)


def test_pretrain_driver_on_gpu_assembled_clips(tmp_path):
    """main_byol.py --dataset synthetic_video: sampler -> cstp_clip_assemble -> the pre-training step, two epochs; the losses
    are finite and the pretext losses start near ln(classes) (labels and clips are wired to the right heads)."""
    import importlib.util
    import os
    from cstp_amd.opts import parse_opts
    spec = importlib.util.spec_from_file_location("main_byol", os.path.join(os.path.dirname(__file__), "..", "main_byol.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    opts = parse_opts(["--dataset", "synthetic_video", "--batch_size", "4", "--sample_duration", "8", "--sample_size", "32",
                       "--model_name", "r21d_byol", "--model_depth", "1", "--n_workers", "0", "--synthetic_len", "16",
                       "--result_path", str(tmp_path), "--task", "loss_com", "--loss_weight", "0.1", "1", "1", "1", "1",
                       "--n_epochs", "2", "--learning_rate", "0.01"])
    mod.main(opts)
    rows = open(str(tmp_path / "synthetic_video" / "loss_com" / "synthetic_video_train_clip8modelr21d_byol1.log")).read().strip().split("\n")
    assert len(rows) == 3
    head = rows[0].split("\t")
    for r in rows[1:]:
        v = dict(zip(head, r.split("\t")))
        assert np.isfinite(float(v["loss"]))
        assert 0.5 < float(v["loss_pred_spa"]) < 4 and 0.5 < float(v["loss_pred_rot"]) < 4
