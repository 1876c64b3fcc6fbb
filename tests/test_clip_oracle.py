"""Pin the data-path oracle (oracle/po.py) bit-for-bit against Pillow -- the reference's image library, present in this
image -- and check the sampler's decisions (cstp_amd/sampler.py) against the invariants the reference code establishes.  CPU."""
import random

import numpy as np
import pytest
from PIL import Image

from oracle import pil_ops as po

PIL_ROT = {90: Image.ROTATE_90, 180: Image.ROTATE_180, 270: Image.ROTATE_270}


@pytest.mark.parametrize("h,w,oh,ow", [(120, 160, 112, 112), (56, 73, 112, 112), (240, 320, 112, 112), (33, 47, 17, 29),
                                       (100, 100, 100, 37), (64, 64, 112, 64), (171, 128, 112, 112), (9, 200, 112, 112),
                                       (112, 112, 112, 112), (3, 3, 8, 8)])
def test_bicubic_resize_is_pillow_bit_exact(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(po.resize_bicubic(img, ow, oh), ref)
    # the product's coefficient tables are the oracle's
    from cstp_amd.clip_ops import resize_tables
    for n_in, n_out in ((w, ow), (h, oh)):
        ks, b, k = resize_tables(n_in, n_out)
        ks2, b2, k2 = po.precompute_coeffs(n_in, 0.0, float(n_in), n_out)
        assert ks == ks2 and np.array_equal(b, b2) and np.array_equal(k, k2)


def test_transposes_crop_and_tensor_match_pillow():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (11, 17, 3), dtype=np.uint8)
    pim = Image.fromarray(img, "RGB")
    for code, pc in PIL_ROT.items():
        assert np.array_equal(po.transpose(img, code), np.asarray(pim.transpose(pc)))
    assert np.array_equal(po.transpose(img, "flip"), np.asarray(pim.transpose(Image.FLIP_LEFT_RIGHT)))
    for box in ((2, 3, 9, 10), (5, 4, 25, 9), (-3, -2, 6, 20), (12, 8, 30, 30)):       # inside, past the right, around, corner
        assert np.array_equal(po.crop(img, box), np.asarray(pim.crop(box)))
    t = po.to_tensor_tf(img)
    assert t.shape == (3, 11, 17) and t.dtype == np.float32
    assert np.array_equal(t, np.clip(img.astype(np.float32).transpose(2, 0, 1) / np.float32(255) * 2 - 1, -1, 1))
    # the whole per-frame chain against PIL calls, as the reference chains them
    frames = rng.integers(0, 256, (4, 60, 80, 3), dtype=np.uint8)
    clip = po.assemble_clip(frames, [3, 1], 270, (5, 7, 45, 62), 32, True)
    ref = []
    for f in (3, 1):
        im = Image.fromarray(frames[f], "RGB").transpose(Image.ROTATE_270).crop((5, 7, 45, 62)).resize((32, 32), Image.BICUBIC)
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
        ref.append(np.asarray(im).astype(np.float32).transpose(2, 0, 1) / np.float32(255) * 2 - 1)
    assert np.array_equal(clip, np.stack(ref, axis=1))


def test_sampler_invariants():
    from cstp_amd import sampler
    seen_tem, seen_spa, seen_pb = set(), set(), set()
    n_base, orders, kinds = 0, set(), set()
    for seed in range(300):
        rng = random.Random(seed)
        total = rng.choice([20, 40, 90, 150, 300])
        plan = sampler.sample_pair(total, 171, 128, 16, rng, np_rng=np.random.RandomState(seed))
        a, b = plan.clip_1, plan.clip_2
        assert len(a.frames) == len(b.frames) == 16
        assert all(0 <= f < total for f in a.frames + b.frames)
        rate = sampler.PACE[plan.pb_label]
        assert plan.pb_label <= min(3, int(np.log2(total / 15)))          # datasets.py:871-872
        assert a.rotate == sampler.ROTATE[plan.rot_labels[0]] and b.rotate == sampler.ROTATE[plan.rot_labels[1]]
        clip_range = 15 * rate
        if total - clip_range <= 0:                                        # short video: wrapped indices, both clips equal
            assert a.frames == b.frames and plan.tem_label == 0
        else:
            assert all(y - x == rate for x, y in zip(a.frames, a.frames[1:]))
            shift = abs(b.frames[0] - a.frames[0])
            assert shift == int((1 - sampler.OVERLAP_TEM_RATE[plan.tem_label]) * clip_range)   # :917-923
        # crops: same size, inside their rotated frames, overlapping by the labelled share of the first crop's area (:516-562)
        r1w, r1h = (128, 171) if a.rotate in (90, 270) else (171, 128)
        assert 0 <= a.box[0] < a.box[2] <= r1w and 0 <= a.box[1] < a.box[3] <= r1h
        assert b.box[0] >= 0 and b.box[1] >= 0                  # the second box may reach past the right / bottom edge when the
        if (a.rotate in (90, 270)) == (b.rotate in (90, 270)):  # rotated sizes differ (:535-541 check two sides); PIL pads with 0
            assert b.box[2] <= r1w and b.box[3] <= r1h
        w1, h1 = a.box[2] - a.box[0], a.box[3] - a.box[1]
        assert (b.box[2] - b.box[0], b.box[3] - b.box[1]) == (w1, h1)
        assert 0.2 * 171 * 128 * 0.98 <= w1 * h1 <= 171 * 128
        ow = max(0, min(a.box[2], b.box[2]) - max(a.box[0], b.box[0]))
        oh = max(0, min(a.box[3], b.box[3]) - max(a.box[1], b.box[1]))
        rate_spa = sampler.OVERLAP_SPA_RATE[plan.spa_label]
        # s_w >= int(rate * p_w) and s_h = int(rate * area / s_w) are floored (s_h can even exceed p_h by one): a row + a column
        assert abs(ow * oh - rate_spa * w1 * h1) <= w1 + h1 + 1
        seen_tem.add(plan.tem_label); seen_spa.add(plan.spa_label); seen_pb.add(plan.pb_label)
        # a plan is a pure function of its arguments and seed
        rng2 = random.Random(seed)
        assert rng2.choice([20, 40, 90, 150, 300]) == total
        assert sampler.sample_pair(total, 171, 128, 16, rng2, np_rng=np.random.RandomState(seed)) == plan
        # the base_transform draws (preprocess_data.py:1110-1119) stay inside the ranges the reference draws them from
        for c in (a, b):
            assert (c.base is not None) == c.use_base
            if c.base is not None:
                n_base += 1
                assert -10 <= c.base.angle <= 10
                if c.base.jitter is not None:
                    assert sorted(o for o, _ in c.base.jitter) == ["brightness", "contrast", "hue", "saturation"]
                    assert all((-0.1 <= f <= 0.1) if o == "hue" else (0.6 <= f <= 1.4) for o, f in c.base.jitter)
                    orders.add(tuple(o for o, _ in c.base.jitter))
                assert c.base.gray is None or (len(c.base.gray) == 16 and set(c.base.gray) <= {0, 1, 2})
                assert c.base.blur_sigma is None or 0.1 <= c.base.blur_sigma <= 2.0
                kinds.update([("jitter", c.base.jitter is not None), ("gray", c.base.gray is not None),
                              ("blur", c.base.blur_sigma is not None)])
    assert seen_tem == {0, 1, 2, 3, 4} and seen_spa == {0, 1, 2, 3, 4} and seen_pb == {0, 1, 2, 3}
    assert 0.2 * 600 < n_base < 0.4 * 600 and len(orders) > 6           # p = 0.3 per clip; the colour operations are shuffled
    assert kinds == {(k, v) for k in ("jitter", "gray", "blur") for v in (True, False)}


def test_sampler_draw_order_matches_the_reference_sequence():
    """The reference consumes the global `random` stream in a fixed order (datasets.py:872-915, preprocess_data.py:493-533,
    578, 726-735, 779); replaying the same order by hand must land on the plan."""
    from cstp_amd import sampler
    seed, total, t = 7, 200, 16
    plan = sampler.sample_pair(total, 171, 128, t, random.Random(seed))
    r = random.Random(seed)
    pb = r.randint(0, min(3, int(np.log2(total / (t - 1)))))
    rot1, rot2 = r.randint(0, 3), r.randint(0, 3)
    assert (pb, (rot1, rot2)) == (plan.pb_label, plan.rot_labels)
    start = r.randint(1, total - 15 * sampler.PACE[pb])
    assert plan.clip_1.frames[0] == start - 1
    # ... and through base_transform (:1110-1119): find a seed whose first clip takes the base branch with every option on, replay
    for seed in range(400):
        plan = sampler.sample_pair(total, 171, 128, t, random.Random(seed), np_rng=np.random.RandomState(seed))
        bp = plan.clip_1.base
        if bp is not None and bp.jitter is not None and bp.gray is not None and bp.blur_sigma is not None:
            break
    else:
        raise AssertionError("no seed takes the full base branch")
    r = random.Random(seed)
    frames = sampler.sample_frames(total, t, r)
    r.choices(range(2), weights=[1, 0])
    assert (r.random() < 0.3) is True
    r.random()                                              # clip 2's base-or-null draw
    crop = sampler.OverlapCrop(r)
    w1, h1 = (128, 171) if plan.clip_1.rotate in (90, 270) else (171, 128)
    assert crop.first(w1, h1) == plan.clip_1.box
    assert r.uniform(-10, 10) == bp.angle                   # RandomRotation
    assert not 0.8 < r.random()                             # RandomApply(ColorJitter, p = 0.8) applies
    r.random()                                              # ClipColorJitter's own p = 1.0 draw
    ops = [("brightness", r.uniform(0.6, 1.4)), ("contrast", r.uniform(0.6, 1.4)), ("saturation", r.uniform(0.6, 1.4)),
           ("hue", r.uniform(-0.1, 0.1))]
    r.shuffle(ops)
    assert ops == bp.jitter
    assert r.random() < 0.2                                 # ClipRandomGray
    assert [int(c) for c in np.random.RandomState(seed).choice(3, size=1)] == bp.gray[:1]
    assert not 0.5 < r.random()                             # RandomApply(GaussianBlur, p = 0.5) applies
    assert r.uniform(0.1, 2.0) == bp.blur_sigma
    assert (r.random() < 0.5) == plan.clip_1.flip
    assert frames[0] == plan.clip_1.frames


def test_gpu_clip_loader_shards_without_a_gpu():
    """GpuClipLoader's index plan is host logic: ranks get disjoint, equal strides of one epoch-seeded permutation."""
    from cstp_amd.clip_ops import GpuClipLoader

    class _DS:
        def __len__(self):
            return 37
    parts = [GpuClipLoader(_DS(), 4, rank=r, world_size=3, seed=5) for r in range(3)]
    for p in parts:
        p.set_epoch(2)
    idx = [p.indices() for p in parts]
    assert [len(i) for i in idx] == [12, 12, 12] and len(set(sum(idx, []))) == 36
    assert [len(p) for p in parts] == [3, 3, 3]
    parts[0].set_epoch(3)
    assert parts[0].indices() != idx[0]
    with pytest.raises(ValueError):
        GpuClipLoader(_DS(), 4, rank=3, world_size=3)


# ---- the base_transform branch (preprocess_data.py:1110-1121): every numpy restatement bit-for-bit against the Pillow call the
#      reference (or torchvision's PIL backend under it) makes -------------------------------------------------------------------
def _imgs():
    rng = np.random.RandomState(7)
    smooth = np.clip(rng.randn(112, 112, 3) * 40 + 128, 0, 255).astype(np.uint8)
    return [rng.randint(0, 256, (112, 112, 3), dtype=np.uint8), smooth, rng.randint(0, 256, (37, 64, 3), dtype=np.uint8)]


def test_small_angle_rotation_matches_pillow():
    """RandomRotation(10) :1060-1100 -> Image.rotate(angle): NEAREST, same size, black corners (Geometry.c affine_fixed)."""
    for img in _imgs():
        for angle in (3.7, -9.99, 10.0, -10.0, 0.001, -0.5, 0.0, 45.0, 90.0, 180.0, 270.0, 359.2):
            ref = np.asarray(Image.fromarray(img).rotate(angle))
            assert np.array_equal(po.rotate_nearest(img, angle), ref), (img.shape, angle)


def test_colour_jitter_ops_match_pillow():
    """ClipColorJitter :584-672 -> torchvision F.adjust_brightness / _contrast / _saturation = ImageEnhance blends,
    F.adjust_hue = HSV round trip with a uint8 hue shift (torchvision/transforms/functional_pil.py)."""
    from PIL import ImageEnhance
    for img in _imgs():
        pim = Image.fromarray(img)
        for f in (0.6, 1.4, 0.0, 1.0, 1.0001, 0.73219, 1.39999):
            assert np.array_equal(po.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pim).enhance(f)))
            assert np.array_equal(po.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pim).enhance(f)))
            assert np.array_equal(po.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pim).enhance(f)))
        for f in (0.1, -0.1, 0.05, -0.0371, 0.5, -0.5, 0.0):
            h, s, v = pim.convert("HSV").split()
            nh = np.array(h, dtype=np.uint8)
            with np.errstate(over="ignore"):
                nh += np.array(f * 255).astype(np.uint8)
            ref = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
            assert np.array_equal(po.adjust_hue(img, f), np.asarray(ref)), f


def test_hsv_conversions_match_pillow_on_every_colour():
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    cube = np.stack([r, g, b], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(po.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube).convert("HSV")))
    assert np.array_equal(po.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))
    assert np.array_equal(po.rgb_to_l(cube), np.asarray(Image.fromarray(cube).convert("L")))


def test_channel_gray_and_gaussian_blur_match_pillow():
    """ClipRandomGray.grayscale :704-709; ClipGaussianBlur :675-687 -> ImageFilter.GaussianBlur(radius = sigma in [0.1, 2])."""
    from PIL import ImageFilter
    for img in _imgs():
        for ch in range(3):
            np_img = np.array(Image.fromarray(img))[:, :, ch]
            ref = np.asarray(Image.fromarray(np.dstack([np_img, np_img, np_img]), "RGB"))
            assert np.array_equal(po.channel_gray(img, ch), ref)
        for rad in list(np.linspace(0.1, 2.0, 39)) + [0.3, 1.0, 1.5, 3.7]:
            ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=float(rad))))
            assert np.array_equal(po.gaussian_blur(img, float(rad)), ref), rad
