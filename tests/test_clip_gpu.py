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
        plan = sampler.sample_pair(70, 128, 96, 16, random.Random(seed), np_rng=np.random.RandomState(seed))
        (c1, c2), labels = clip_ops.assemble_pair(dev, plan, 64)
        for c, p in ((c1, plan.clip_1), (c2, plan.clip_2)):
            if p.base is None:          # null_transform
                want = po.assemble_clip(frames, p.frames, p.rotate, p.box, 64, p.flip)
            else:                       # base_transform on the resized 8-bit frames
                u8 = np.stack([po.resize_bicubic(po.crop(po.transpose(frames[f], p.rotate), p.box), 64, 64) for f in p.frames])
                want = po.base_transform_clip(u8, p.base, p.flip)
            assert np.array_equal(c.cpu().numpy(), want)
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


# ---- the base_transform branch (preprocess_data.py:1110-1121): every GPU kernel np.array_equal to the Pillow call the reference makes
def _clips():
    g = np.random.default_rng(11)
    smooth = np.clip(g.normal(128, 40, (3, 112, 112, 3)), 0, 255).astype(np.uint8)
    return [g.integers(0, 256, (4, 112, 112, 3), dtype=np.uint8), smooth, g.integers(0, 256, (2, 37, 64, 3), dtype=np.uint8)]


def _pil_frames(clip, fn):
    return np.stack([np.asarray(fn(Image.fromarray(f, "RGB"))) for f in clip])


def test_small_angle_rotation_is_pillow_bit_exact():
    from cstp_amd import clip_ops
    for clip in _clips():
        dev = torch.from_numpy(clip).cuda()
        for angle in (3.7, -9.99, 10.0, -10.0, 0.001, -0.5, 0.0, 45.0, 90.0, 180.0, 270.0, 359.2):
            ref = _pil_frames(clip, lambda im: im.rotate(angle))
            assert np.array_equal(clip_ops.clip_rotate(dev, angle).cpu().numpy(), ref), (clip.shape, angle)


def test_colour_jitter_ops_are_pillow_bit_exact():
    """adjust_brightness / _contrast / _saturation as torchvision's PIL backend performs them (ImageEnhance blends), adjust_hue
    as its HSV round trip (functional_pil.py); the chains of tests/test_clip_oracle.py."""
    from PIL import ImageEnhance
    from cstp_amd import clip_ops
    enh = {"brightness": ImageEnhance.Brightness, "contrast": ImageEnhance.Contrast, "saturation": ImageEnhance.Color}

    def tv_hue(im, f):
        h, s, v = im.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh += np.array(f * 255).astype(np.uint8)
        return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    for clip in _clips():
        dev = torch.from_numpy(clip).cuda()
        for op, cls in enh.items():
            for f in (0.6, 1.4, 0.0, 1.0, 1.0001, 0.73219, 1.39999):
                ref = _pil_frames(clip, lambda im: cls(im).enhance(f))
                assert np.array_equal(clip_ops.clip_colour(dev, op, f).cpu().numpy(), ref), (op, f)
        for f in (0.1, -0.1, 0.05, -0.0371, 0.5, -0.5, 0.0):
            ref = _pil_frames(clip, lambda im: tv_hue(im, f))
            assert np.array_equal(clip_ops.clip_colour(dev, "hue", f).cpu().numpy(), ref), f


def test_hsv_conversions_are_pillow_bit_exact_on_every_colour():
    """All 2^24 RGB triples through RGB -> HSV and all 2^24 HSV triples through HSV -> RGB (Convert.c's float / double mix)."""
    import ctypes
    from cstp_amd import _lib
    lib = _lib.load()
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    cube = np.stack([r, g, b], -1).astype(np.uint8).reshape(4096, 4096, 3)
    dev = torch.from_numpy(cube).cuda()
    out = torch.empty_like(dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cstp_clip_hue(st, dev.data_ptr(), out.data_ptr(), 4096 * 4096, 0, 1), "cstp_clip_hue")
    assert np.array_equal(out.cpu().numpy(), np.asarray(Image.fromarray(cube).convert("HSV")))
    _lib.check(lib.cstp_clip_hue(st, dev.data_ptr(), out.data_ptr(), 4096 * 4096, 0, 2), "cstp_clip_hue")
    assert np.array_equal(out.cpu().numpy(), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


def test_channel_gray_and_gaussian_blur_are_pillow_bit_exact():
    from PIL import ImageFilter
    from cstp_amd import clip_ops
    for clip in _clips():
        dev = torch.from_numpy(clip).cuda()
        chans = [i % 3 for i in range(clip.shape[0])]
        ref = np.stack([np.repeat(f[:, :, c:c + 1], 3, axis=2) for f, c in zip(clip, chans)])
        assert np.array_equal(clip_ops.clip_gray(dev, chans).cpu().numpy(), ref)
        for sigma in list(np.linspace(0.1, 2.0, 20)) + [0.3, 1.0, 1.5, 3.7]:
            ref = _pil_frames(clip, lambda im: im.filter(ImageFilter.GaussianBlur(radius=float(sigma))))
            assert np.array_equal(clip_ops.clip_gaussian_blur(dev, float(sigma)).cpu().numpy(), ref), sigma
            r_int, ww, fw = clip_ops.gaussian_box_weights(float(sigma))
            assert r_int == int(po.gaussian_box_radius(float(sigma)))          # the host constants are the oracle's


def test_base_transform_clips_match_pillow_chain_and_oracle():
    """Whole clips through assemble_clip with base_transform draws (cstp_amd.sampler.base_draws): identical to the Pillow calls
    chained as the reference chains them (crop -> resize, then rotate -> colour jitter in the drawn order -> channel gray ->
    Gaussian blur -> flip -> tensor), and to the numpy oracle."""
    from PIL import ImageEnhance, ImageFilter
    from cstp_amd import clip_ops, sampler
    enh = {"brightness": ImageEnhance.Brightness, "contrast": ImageEnhance.Contrast, "saturation": ImageEnhance.Color}
    g = np.random.default_rng(3)
    frames = g.integers(0, 256, (24, 96, 128, 3), dtype=np.uint8)
    dev = torch.from_numpy(frames).cuda()
    n_base = 0
    for seed in range(40):
        plan = sampler.sample_pair(24, 128, 96, 8, random.Random(seed), p_base=0.7, np_rng=np.random.RandomState(seed))
        for cp in (plan.clip_1, plan.clip_2):
            if cp.base is None:
                continue
            n_base += 1
            got = clip_ops.assemble_clip(dev, cp, 112).cpu().numpy()
            out = []
            for i, f in enumerate(cp.frames):
                im = Image.fromarray(frames[f], "RGB")
                if cp.rotate:
                    im = im.transpose(PIL_ROT[cp.rotate])
                im = im.crop(cp.box).resize((112, 112), Image.BICUBIC)
                im = im.rotate(cp.base.angle)
                for op, fac in (cp.base.jitter or ()):
                    if op == "hue":
                        h, s, v = im.convert("HSV").split()
                        nh = np.array(h, dtype=np.uint8)
                        with np.errstate(over="ignore"):
                            nh += np.array(fac * 255).astype(np.uint8)
                        im = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
                    else:
                        im = enh[op](im).enhance(fac)
                if cp.base.gray is not None:
                    c = np.array(im)[:, :, cp.base.gray[i]]
                    im = Image.fromarray(np.dstack([c, c, c]), "RGB")
                if cp.base.blur_sigma is not None:
                    im = im.filter(ImageFilter.GaussianBlur(radius=cp.base.blur_sigma))
                if cp.flip:
                    im = im.transpose(Image.FLIP_LEFT_RIGHT)
                out.append(po.to_tensor_tf(np.asarray(im)))
            ref = np.stack(out, axis=1)
            assert np.array_equal(got, ref), (seed, cp.base)
            # ... and the oracle's own chain
            u8 = np.stack([po.resize_bicubic(po.crop(po.transpose(frames[f], cp.rotate), cp.box), 112, 112) for f in cp.frames])
            assert np.array_equal(po.base_transform_clip(u8, cp.base, cp.flip), ref)
    assert n_base >= 20
