#!/usr/bin/env python3
"""Headline benchmark: CSTP pre-training clips/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one full optimisation step of the hot path on one resident synthetic batch:
4 encoder forwards (2 online + 2 target), EMA, heads, losses, 2 encoder backwards, DDP gradient
all-reduce (N > 1), clip_grad_norm_(18), SGD -- plus the driver's per-iteration read of the loss scalars
(one pinned-memory copy, read one step late: cstp_amd.train.LaggedScalars, as main_byol.py does for its
log line).  One "clip" = one (clip_1, clip_2) pair.

Workload (BASELINE.json configs[1]): R(2+1)D-18, B=16 per GPU, 3x16x112x112 fp32, BYOL + NT-Xent
(negatives all-gathered across ranks) + overlap-rate heads (loss_weight 0.1 1 1 0 0, ntxent 1);
weak scaling: per-GPU batch fixed, global batch = 16 N.

Prints ONE JSON line on rank 0.  Besides the contract's fields:
  * ``roofline``   -- the dominant kernel (the S1-shape spatial conv forward), timed with HIP events on its launch
                      stream INSIDE the timed region; ``roofline.kernels`` -- the same measurement for the other heavy
                      kernels (S1 data / weight gradient, T1 forward, the 3-channel stem, BatchNorm forward / backward),
                      taken in two extra steps after the timed region with the stream overlaps switched off so that an
                      event pair brackets one kernel chain running alone; each row carries its algorithmic FLOPs and
                      bytes, its fraction of the compute ceiling of the arithmetic it ran and of the HBM peak;
  * ``arithmetics`` -- clips/s of the same step under the three GEMM arithmetics the library has, measured in this
                      process (N = 1 only): the default f16 pair (= ``value``), the exact bf16 triple, native f32 MFMA;
  * ``cpu_baseline`` -- the CPU oracle timed on this host's cores on a bounded sample (1 warm-up + 3 timed steps;
                      rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 achievable
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
BF16_MFMA_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: BF16/F16 dense (~2.5 PF)
SPLIT_PRODUCTS = {2: 3, 3: 6}    # 16-bit MFMA products per fp32 product: f16 pair / bf16 triple (csrc/igemm_split.h)
ARITH_NAME = {0: "native f32 MFMA", 1: "native f32 MFMA", 2: "2xf16-split (22-bit operands), f16 MFMA x3, f32 accumulate",
              3: "3xbf16-split (exact), bf16 MFMA x6, f32 accumulate"}
DEPTH, B_LOCAL, T, HW = 18, 16, 16, 112
LOSS_WEIGHT = (0.1, 1.0, 1.0, 0.0, 0.0)
NTXENT_WEIGHT = 1.0
PMC_FILE = os.path.join(ROOT, "profiles", "r04", "pmc_dominant_kernel.json")


class KernelTimers:
    """HIP-event timers for named ops: ``specs`` = {name: (what, key)} as cstp_amd.ops._span reports them.  A pair of events
    is recorded on the CURRENT stream (= the stream the C ABI launches on) around each matching call while ``enabled``
    holds the name; ``main_only`` restricts to launches on the step's main stream (a side stream overlaps its kernels
    with the main stream's: an event pair around such a launch measures queueing, not the kernel)."""

    def __init__(self, specs, main_stream):
        self.by_key = {(what, key): name for name, (what, key) in specs.items()}
        self.pairs = {name: [] for name in specs}
        self.enabled = set()
        self.main_only = True
        self.main_stream = main_stream

    class _Span:
        def __init__(self, sink):
            self.sink = sink

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
            return self

        def __exit__(self, *exc):
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.sink.append((self.a, b))
            return False

    def span(self, what, key):
        name = self.by_key.get((what, key))
        if name is None or name not in self.enabled:
            return None
        if self.main_only and torch.cuda.current_stream().cuda_stream != self.main_stream:
            return None
        return KernelTimers._Span(self.pairs[name])

    def mean_ms(self, name):
        p = self.pairs[name]
        return sum(a.elapsed_time(b) for a, b in p) / len(p) if p else 0.0


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def pmc_traffic(kernel_tile):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 x2 read correction) -- a profiler measurement cannot be taken inside this process, so the
    number travels with the profile, stamped with the kernel variant and commit it was measured on; it is reported only
    while the library still runs that variant for the layer (else None)."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        if list(rec["tile"]) != list(kernel_tile):
            return None, {"note": "committed PMC record is for tile %r, the library now runs %r" % (rec["tile"], list(kernel_tile))}
        return rec["hbm_bytes_per_launch"], {"source": os.path.relpath(PMC_FILE, ROOT), "kernel": rec.get("kernel"),
                                             "measured_at_commit": rec.get("commit")}
    except (OSError, KeyError, ValueError):
        return None, {"note": "no committed PMC record"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_b=2, timed=3):
    """The CPU oracle (stock PyTorch CPU ops; `port`) on a bounded sample of the same workload: 1 warm-up step, then
    ``timed`` full steps."""
    from oracle import r21d_byol_oracle as orc
    ls = orc.layer_sizes_for_depth(DEPTH)
    cores = torch.get_num_threads()
    sd = orc.closed_form_state(ls, torch.float32)
    mom = {}
    x1, x2, labels = orc.closed_form_clips(sample_b, T, HW, torch.float32)
    # the SAME objective as the GPU line: loss_weight (0.1, 1, 1, 0, 0) + 1 x NT-Xent on the online projections (configs[1])
    kw = dict(ntxent_weight=NTXENT_WEIGHT, temperature=0.5)
    orc.train_step(sd, mom, x1, x2, labels, ls, 0.01, 0.9, 5e-4, LOSS_WEIGHT, True, **kw)          # warm-up
    t0 = time.time()
    for _ in range(timed):
        orc.train_step(sd, mom, x1, x2, labels, ls, 0.01, 0.9, 5e-4, LOSS_WEIGHT, True, **kw)
    dt = (time.time() - t0) / timed
    return {"value": sample_b / dt, "unit": "clips/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": "R(2+1)D-18 full step (fwd+bwd+clip+SGD+EMA; the GPU line's objective: BYOL + overlap heads + NT-Xent), "
                      "B=%d of %d clip pairs 3x%dx%dx%d, 1 warm-up + %d timed steps, %.1f s per step"
                      % (sample_b, B_LOCAL, T, HW, HW, timed, dt)}


def conv_work(n, c, d, h, w, k, kt, kh, kw, st, sh, sw, pt, ph, pw):
    """Algorithmic FLOPs and bytes of one convolution call (2 MACs; every tensor touched once)."""
    do, ho, wo = (d + 2 * pt - kt) // st + 1, (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    flops = 2.0 * k * c * kt * kh * kw * n * do * ho * wo
    nbytes = 4.0 * (n * c * d * h * w + n * k * do * ho * wo + k * c * kt * kh * kw)
    return flops, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel table and the other two arithmetics")
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--batch", type=int, default=B_LOCAL)
    ap.add_argument("--frames", type=int, default=T, help="clip length (BASELINE configs[3] uses 32)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # under torch.distributed.run
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        world, rank, local_rank, launched = 1, 0, 0, False
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under the launcher the data-parallel machinery (RCCL process group, BN-buffer broadcast, flat gradient all-reduce,
    # NT-Xent all-gather) runs even at world size 1, so the N = 1 launcher run exercises the N > 1 code path.
    if launched:
        dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank)

    from cstp_amd import _lib, ops, r21d_byol
    from cstp_amd.ntxent import NTXentLoss
    from cstp_amd.optim import FlatSGD
    from cstp_amd.r21d_byol import R21DBYOL, layer_sizes_for_depth
    from cstp_amd.synthetic import device_batch
    from cstp_amd.train import LaggedScalars, PretrainStep

    torch.manual_seed(1)                       # opts.py:160 default seed; random-init weights
    model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(args.depth))
    model.cuda(local_rank)
    arenas = model.flatten_parameters()
    model.train()
    ddp = model
    if launched:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                        find_unused_parameters=False)
    opt = FlatSGD(model.parameters(), lr=0.09, momentum=0.9, weight_decay=5e-4, arenas=arenas)
    ntx = NTXentLoss(device=dev, batch_size=args.batch * world, temperature=0.5, use_cosine_similarity=True)
    step = PretrainStep(ddp, opt, LOSS_WEIGHT, clip_grad_norm=True, ntxent=ntx, ntxent_weight=NTXENT_WEIGHT)
    x1, x2, lab = device_batch(args.batch, args.frames, HW, dev, seed=1 + rank)
    lagged = LaggedScalars(dev, world)

    # both views of the pair share one launch (batch 2B, two BN groups)
    nb, fr, h2 = 2 * args.batch, args.frames, HW // 2
    d_s1 = (nb, 64, fr, h2, h2, 144, 1, 3, 3, 1, 1, 1, 0, 1, 1)          # S1: 64 -> 144, 1x3x3 at 16x56x56
    d_t1 = (nb, 144, fr, h2, h2, 64, 3, 1, 1, 1, 1, 1, 1, 0, 0)          # T1: 144 -> 64, 3x1x1
    d_s0 = (nb, 3, fr, HW, HW, 83, 1, 7, 7, 1, 2, 2, 0, 3, 3)            # stem S0: 3 -> 83, 1x7x7 stride 2
    bn_key = (nb, 144, fr * h2 * h2, 2, False, True)                     # BN + ReLU behind S1 (two view groups)
    specs = {"S1 fwd": ("conv3d_forward", d_s1), "S1 dgrad": ("conv3d_backward_data", d_s1),
             "S1 wgrad": ("conv3d_backward_weight", d_s1), "T1 fwd": ("conv3d_forward", d_t1),
             "T1 dgrad": ("conv3d_backward_data", d_t1), "T1 wgrad": ("conv3d_backward_weight", d_t1),
             "stem S0 fwd": ("conv3d_forward", d_s0), "stem S0 wgrad": ("conv3d_backward_weight", d_s0),
             "BN+ReLU fwd (144 ch)": ("bn_forward", bn_key), "BN+ReLU bwd (144 ch)": ("bn_backward", bn_key)}
    timers = KernelTimers(specs, torch.cuda.current_stream().cuda_stream)
    ops.kernel_timer = timers

    def run(n):
        for _ in range(n):
            out = step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
            lagged.push(out)                     # the driver's per-iteration log read: no host sync (one step late)
        lagged.flush()

    def timed_loop(n):
        if launched:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        if launched:
            dist.barrier()
        el = time.perf_counter() - t0
        if launched:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        return el

    run(1)                # untimed: first sight of every layer geometry reads (or times) its tile
    run(args.warmup)
    timers.enabled = {"S1 fwd"}
    elapsed = timed_loop(args.steps)
    timers.enabled = set()

    lib = _lib.load()
    default_terms = lib.cstp_gemm_get_split_terms()

    def query(desc_t, mode):
        tile = (ctypes.c_int32 * 4)()
        _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(_lib.ConvDesc(*desc_t)), mode, tile), "cstp_conv3d_query_tile")
        return [int(v) for v in tile]

    extras = rank == 0 and world == 1 and not args.no_extras
    kernels, arith = [], None
    same_cfg = args.batch == B_LOCAL and args.depth == DEPTH and args.frames == T
    if extras:
        # ---- per-kernel table: two more steps with the stream overlaps off, every timer on
        ov_w, ov_t = ops.OVERLAP_WGRAD, r21d_byol.OVERLAP_TARGET_FORWARD
        ops.OVERLAP_WGRAD, r21d_byol.OVERLAP_TARGET_FORWARD = False, False
        dom_pairs = timers.pairs["S1 fwd"]
        timers.pairs["S1 fwd"] = []
        timers.enabled = set(specs)
        run(2)
        torch.cuda.synchronize()
        timers.enabled = set()
        ops.OVERLAP_WGRAD, r21d_byol.OVERLAP_TARGET_FORWARD = ov_w, ov_t
        mode_of = {"conv3d_forward": 0, "conv3d_backward_data": 1, "conv3d_backward_weight": 2}
        for name, (what, key) in specs.items():
            ms = timers.mean_ms(name)
            row = {"kernel": name, "launches_timed": len(timers.pairs[name]), "avg_ms": ms}
            if what.startswith("conv3d"):
                flops, nbytes = conv_work(*key)
                from cstp_amd import r21d_byol as _rb2
                # the temporal layer's forward carries the BatchNorm + ReLU in front as an in_affine (mode 3: its own variant)
                t1_aff = name == "T1 fwd" and _rb2.FUSE_BN_TEMPORAL and \
                    ops.in_affine_fused(d_t1[:5], (d_t1[5], d_t1[1]) + tuple(d_t1[6:9]), d_t1[9:12], d_t1[12:15], 2)
                tile = query(key, 3 if t1_aff else mode_of[what])
                terms = tile[2]
                peak = BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS[terms] if terms in SPLIT_PRODUCTS else F32_MFMA_PEAK_TFLOPS
                row.update({"bound": "mfma", "tile": "%dx%d" % (tile[0], tile[1]), "arithmetic": ARITH_NAME[terms],
                            "algorithmic_gflop": flops / 1e9, "algorithmic_bytes": nbytes,
                            "tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "compute_peak_tflops": peak})
                row["frac_compute"] = row["tflops"] / peak
            else:
                n, c, s = key[0], key[1], key[2]
                # the BN behind S1: its statistics come from the convolution's epilogue when that layer runs the patch kernel
                pre = what == "bn_forward" and ops.FUSE_BN_STATS and \
                    lib.cstp_conv3d_bnstats_nsplit(ctypes.byref(ops.ConvDesc(*d_s1)), 2) > 0
                per_elem = (8.0 if pre else 12.0) if what == "bn_forward" else 20.0      # bwd: (x, dy) twice + dx
                nbytes = per_elem * n * c * s
                # ... and since round 3 its apply + ReLU runs inside the temporal convolution's gather (the T1 fwd row carries it):
                # what is left under this span is cstp_bn_finalize_pre, one wave per channel over the partial sums
                from cstp_amd import r21d_byol as _rb
                if what == "bn_forward" and pre and _rb.FUSE_BN_TEMPORAL and \
                        ops.in_affine_fused(d_t1[:5], (d_t1[5], d_t1[1]) + tuple(d_t1[6:9]), d_t1[9:12], d_t1[12:15], 2):
                    row.update({"bound": "latency", "algorithmic_bytes": 0.0,
                                "bytes_note": "statistics folded from the producing convolution's partial sums (cstp_bn_finalize_pre); "
                                              "the apply + ReLU is part of the T1 fwd row (in_affine): the normalised tensor is never written"})
                    row["hbm_tbs"] = 0.0
                    row["frac_hbm"] = 0.0
                    kernels.append(row)
                    continue
                row.update({"bound": "hbm", "algorithmic_bytes": nbytes,
                            "bytes_note": "%d B/element: %s" % (per_elem, ("statistics from the producing convolution's epilogue; "
                                          "apply pass reads x and writes y" if pre else "statistics pass reads x, apply pass reads "
                                          "x and writes y") if what == "bn_forward" else "reduction pass reads x and dy, apply pass "
                                          "reads x and dy and writes dx (ReLU mask recomputed from x)")})
            row["hbm_tbs"] = nbytes / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            row["frac_hbm"] = row["hbm_tbs"] / HBM_PEAK_TBS
            if what.startswith("conv3d") and row["frac_hbm"] > row["frac_compute"]:
                row["bound"] = "hbm"           # (the temporal layers: 66 FLOP per byte, below the f16-pair ridge of 105)
            kernels.append(row)
        timers.pairs["S1 fwd"] = dom_pairs
        # ---- the other two arithmetics, same process, same step (each has its own class of tuned tiles)
        arith = {"f16x2" if default_terms == 2 else ARITH_NAME[default_terms]: args.batch * args.steps / elapsed}
        for label, terms in (("bf16x3", 3), ("f32_native", 1)):
            if terms == default_terms:
                continue
            ops.set_split_terms(terms)
            run(2)                                                # tiles of this arithmetic + one warm step
            el = timed_loop(3)
            arith[label] = args.batch * 3 / el
        ops.set_split_terms(0)
        arith["note"] = ("clips/s of the identical step in this process; `value` is the first entry; 3 timed steps each for "
                         "the other two; f32_native = every GEMM on v_mfma_f32 (bit-for-bit an fmaf chain), bf16x3 = exact "
                         "3-term bf16 split (6 MFMA products per fp32 product)")

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        clips_s = args.batch * world * args.steps / elapsed
        k_ms = timers.mean_ms("S1 fwd")
        flops, alg_bytes = conv_work(*d_s1)
        ach = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        tile = query(d_s1, 0)
        terms = tile[2]                               # 0 native f32 MFMA, 2 = f16 pair, 3 = bf16 triple
        split = terms in SPLIT_PRODUCTS
        products = SPLIT_PRODUCTS.get(terms, 1)
        # fp32-equivalent peak of the kernel that ran: native f32 MFMA 157.3 TF/s; a split kernel issues 3 (f16 pair) or 6
        # (bf16 triple) 16-bit MFMA products per fp32 product, so its ceiling is the 16-bit dense peak / that count (the
        # algorithmic FLOPs stay the fp32 ones)
        peak = BF16_MFMA_PEAK_TFLOPS / products if split else F32_MFMA_PEAK_TFLOPS
        traffic, traffic_src = pmc_traffic(tile) if same_cfg else (None, {"note": "not the cfg2 workload"})
        line = {
            "metric": "pretrain clips/sec (16x112x112)", "value": clips_s, "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 storage; f16x2-split MFMA (22-bit operands)" if default_terms == 2 else
                      "f32 storage; " + ARITH_NAME[default_terms]),
            "data": "synthetic",
            "arithmetic": ("fp32 storage and f32 accumulation throughout.  GEMM-shaped kernels are chosen per layer geometry "
                           "(persisted table cstp_amd/tuned/) between the native f32 MFMA and the split kernels: every fp32 "
                           "operand scaled by a power of two (per weight row / per activation tensor) and split into an f16 "
                           "pair = 22 significand bits, 3 f16-MFMA products per fp32 product -- NOT IEEE fp32 operands, but "
                           "2e-7..8e-7 rms from fp64 per convolution vs 2e-7..1.3e-6 for the native f32 MFMA chain "
                           "(profiles/r01/split_accuracy.txt).  Parity bar: outputs within 1e-4 (max-abs-diff / max-abs-ref) "
                           "of the reference run in fp64, incl. fixtures with magnitudes spread over six decades inside a "
                           "tensor (tests/golden/*_heavy.npz); the R(2+1)D-34 fixtures use 3e-4 because stock PyTorch fp32 "
                           "itself sits 0.9e-4 from that truth on their projector outputs (tests/test_oracle_golden.py).  "
                           "`arithmetics` holds this step's clips/s under the exact bf16 triple and the native f32 MFMA too"),
            "config": {"workload": "r21d_byol R(2+1)D-%d, B=%d clip pairs/GPU 3x%dx%dx%d, BYOL + NT-Xent(all-gather) + "
                                   "overlap-rate heads, loss_weight 0.1 1 1 0 0, clip 18, SGD; random-init weights"
                                   % (args.depth, args.batch, args.frames, HW, HW),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak,
                         "peak_note": ("16-bit MFMA dense 2516.6 TF/s / %d MFMA products per fp32 product" % products if split
                                       else "f32 MFMA dense"),
                         "frac_of_native_f32_mfma_peak": ach / F32_MFMA_PEAK_TFLOPS,
                         "frac_hbm": alg_bytes / (k_ms * 1e-3) / 1e12 / HBM_PEAK_TBS if k_ms > 0 else 0.0,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel": "spatial conv S1 64->144 1x3x3 @%dx56x56, 2B=%d clips/launch; %dx%d tile, %s; HIP events "
                                   "around the C-ABI call (its weight pack runs with all others of the step from one launch at the step's top: "
                                   "ops.PackPlan)" % (fr, nb, tile[0], tile[1], ARITH_NAME[terms]),
                         "launches_timed": len(timers.pairs["S1 fwd"]), "avg_launch_ms": k_ms,
                         "algorithmic_gflop_per_launch": flops / 1e9,
                         "kernels": kernels,
                         "kernels_note": ("per-kernel rows: HIP events around each C-ABI call (incl. its operand packs / "
                                          "unpacks) in two extra steps after the timed region with weight-gradient and "
                                          "target-forward stream overlaps OFF, so each chain runs alone; compute ceilings "
                                          "2516.6/3, 2516.6/6 or 157.3 TFLOP/s by arithmetic; HBM peak 8 TB/s")},
            "tuned_tiles": dict(ops.tune_stats, table=(os.path.relpath(ops.TUNE_TABLE_PATH, ROOT) if ops.TUNE_TABLE_PATH else None)),
            "commit": git_head(),
        }
        if arith is not None:
            line["arithmetics"] = arith
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
        if os.environ.get("CSTP_DEBUG"):
            print("absmax cells: %r" % (ops.absmax_stats,), file=sys.stderr)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
