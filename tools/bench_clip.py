#!/usr/bin/env python3
"""Throughput of the GPU clip path: sampler (host) + cstp_clip_assemble for one cfg2 batch (16 pairs of 16 x 112 x 112 clips)
from 128x171 (UCF-style short side 128) and 240x320 decoded frames.  Prints one JSON line per frame size."""
import json
import random
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from cstp_amd import clip_ops, sampler  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for h, w in ((128, 171), (240, 320)):
        ds = clip_ops.GpuVideoClips(dev, n_videos=4, frames=120, height=h, width=w, length=4096)
        for i in range(0, 64, 16):
            ds.batch(list(range(i, i + 16)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rng = random.Random(0)
        for _ in range(2000):
            sampler.sample_pair(120, w, h, 16, rng)
        t_plan = (time.perf_counter() - t0) / 2000
        n = 20
        t0 = time.perf_counter()
        for b in range(n):
            ds.batch(list(range(64 + b * 16, 80 + b * 16)))
        torch.cuda.synchronize()
        t_batch = (time.perf_counter() - t0) / n
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plans = [sampler.sample_pair(120, w, h, 16, rng) for _ in range(16)]
        clip_ops.assemble_pair(ds.videos[0], plans[0], 112)
        ev0.record()
        for p in plans:
            clip_ops.assemble_pair(ds.videos[0], p, 112)
        ev1.record()
        torch.cuda.synchronize()
        print(json.dumps({"frames": "%dx%d" % (h, w), "plan_us_per_pair": round(t_plan * 1e6, 1),
                          "batch16_wall_ms": round(t_batch * 1e3, 2), "batch16_gpu_ms": round(ev0.elapsed_time(ev1), 2),
                          "pairs_per_s_wall": round(16 / t_batch, 1)}))


if __name__ == "__main__":
    main()
