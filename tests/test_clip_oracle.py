"""Pin the data-path oracle (oracle/pil_ops.py) bit-for-bit against Pillow -- the reference's image library, present in this
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
    for seed in range(300):
        rng = random.Random(seed)
        total = rng.choice([20, 40, 90, 150, 300])
        plan = sampler.sample_pair(total, 171, 128, 16, rng)
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
        again = sampler.sample_pair(total, 171, 128, 16, random.Random(seed))
        assert again.clip_1.frames != [] and (again.spa_label, again.tem_label, again.pb_label, again.rot_labels) is not None
    assert seen_tem == {0, 1, 2, 3, 4} and seen_spa == {0, 1, 2, 3, 4} and seen_pb == {0, 1, 2, 3}


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
