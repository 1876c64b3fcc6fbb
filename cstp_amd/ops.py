"""torch.autograd wrappers over the C ABI (libcstp_hip.so).  PyTorch supplies device memory,
streams and the autograd graph; every FLOP/byte of the hot path runs in the HIP kernels.

There is deliberately NO CPU implementation here: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, InAffine, check

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

_ws_cache = {}

# Optional per-kernel timers installed by bench.py: an object with ``span(what, key) -> context manager | None`` that brackets
# the C-ABI call of matching ops with HIP events on the launch stream; None in normal operation.
#   what: "conv3d_forward" | "conv3d_backward_data" | "conv3d_backward_weight" (key = the ConvDesc fields as a tuple),
#         "bn_forward" | "bn_backward" (key = (n, c, s, groups, has_residual, relu))
kernel_timer = None


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOSPAN = _NoSpan()


def _span(what, key):
    tm = kernel_timer
    if tm is None:
        return _NOSPAN
    return tm.span(what, key() if callable(key) else key) or _NOSPAN


def _desc_key(desc):
    return tuple(getattr(desc, f) for f, _ in ConvDesc._fields_)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per-device scratch arena (grown geometrically, never shrunk).  All kernels of one op are
    enqueued on the current stream, so one arena per (device, stream) is race-free."""
    key = (device.index, _stream())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        newn = max(nbytes, 1 << 20)
        if ws is not None:
            newn = max(newn, 2 * ws.numel())
        ws = torch.empty(newn, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class PackPlan:
    """The weight packs of a training step from THREE launches instead of ~137 (include/cstp_hip.h: cstp_pack_*).

    Every convolution / linear call re-lays its weights out for its kernel variant (~10 us per call on the step's critical
    chain: 1.3 ms of a 54 ms R(2+1)D-18 step, profiles/r04).  Inside a training step the weights change at exactly two points --
    the optimizer step (online network, predictor, heads) and the EMA (target network) -- so the step that owns the model
    (PretrainStep) hoists the packs: each (weight, direction) call site gets a PERSISTENT workspace; one step RECORDS the pack
    launches of every call (state "record"); from then on ``replay("online")`` at the top of a step and ``replay("target")``
    right behind the EMA run all recorded packs of that group from one launch each, and the calls skip theirs.  Outside an
    armed step (validation, fine-tuning, tests that call the model directly) nothing changes: calls pack for themselves.
    A call whose (weight, tag, descriptor) was not recorded packs for itself too."""

    def __init__(self, target_range=None):
        self.state = "off"            # "off" | "record" | "replay"
        self.armed = False            # True only inside the owning step
        self.ws = {}                  # key -> persistent workspace
        self.recs = {}                # key -> [PackRec, ...] (what that call packs)
        self.tables = {}              # group -> (recs_dev, first_dev, n, total_blocks)
        self.target_range = target_range      # (ptr_lo, ptr_hi) of the EMA target arena: its packs replay behind the EMA
        self.stats = {"replays": 0, "skipped_calls": 0, "recorded_calls": 0}

    @staticmethod
    def key(w_param, tag, x_shape):
        return (w_param.data_ptr(), tag, tuple(x_shape))      # (a weight tensor fixes kernel size, stride and padding)

    def workspace(self, key, device, nbytes):
        ws = self.ws.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None and self.state == "replay":
                self.invalidate()             # a recorded buffer is being replaced: the device tables point at the old one
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self.ws[key] = ws
            self.recs.pop(key, None)          # a new buffer: whatever was recorded for the old one is void
        return ws

    def call(self, key, fn):
        """Run the C-ABI call ``fn`` under the pack mode this key is in."""
        lib = _lib.load()
        if self.state == "record":
            lib.cstp_pack_mode(1)
            try:
                fn()
            finally:
                lib.cstp_pack_mode(0)
                n = lib.cstp_pack_recorded(None, 0)
                buf = (_lib.PackRec * max(n, 1))()
                lib.cstp_pack_recorded(buf, n)
            self.recs[key] = [buf[i] for i in range(n)]
            self.stats["recorded_calls"] += 1
        else:
            # (replay state: the call's workspace is REGISTERED with the library -- cstp_pack_register -- so the call skips its pack
            #  without any per-call switch; the host spends one dictionary lookup here)
            fn()
            if self.state == "replay" and key in self.recs:
                self.stats["skipped_calls"] += 1

    def finish_record(self, device):
        """Build the per-group device tables from what the recording step collected."""
        # three groups: the target network's packs (replayed behind the EMA), the online / predictor / head FORWARD packs (top of
        # the step, main stream) and their DATA-GRADIENT packs (mirrored taps, gathered across the output channels: the expensive
        # ones -- top of the step too, but on the side stream: nothing needs them before the backward pass)
        groups = {"online": [], "online_d": [], "target": []}
        lo, hi = self.target_range if self.target_range is not None else (0, 0)
        for key, recs in self.recs.items():
            for r in recs:
                if lo <= (r.w or 0) < hi:
                    groups["target"].append(r)
                else:
                    groups["online_d" if key[1] == "d" else "online"].append(r)
        self.tables = {}
        for name, recs in groups.items():
            if not recs:
                continue
            arr = (_lib.PackRec * len(recs))(*recs)
            first, tot = [], 0
            for r in recs:
                first.append(tot)
                tot += int(r.nblocks)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self.tables[name] = (raw.to(device), torch.tensor(first, dtype=torch.int32, device=device), len(recs), tot)
        dsts = sorted({int(r.dst) for recs in groups.values() for r in recs if r.dst})
        if dsts:
            arr = (ctypes.c_void_p * len(dsts))(*dsts)
            check(_lib.load().cstp_pack_register(arr, len(dsts), 1), "cstp_pack_register")
        self._registered = dsts
        self.state = "replay"

    def invalidate(self):
        """The kernels a recorded call would run may have changed (arithmetic, deterministic mode, a pinned tile): forget the
        records; the owning step records again after two quiet steps."""
        dsts = getattr(self, "_registered", None)
        if dsts:
            arr = (ctypes.c_void_p * len(dsts))(*dsts)
            _lib.load().cstp_pack_register(arr, len(dsts), 0)
        self._registered = []
        self.state, self.recs, self.tables, self.quiet = "off", {}, {}, 0

    def __del__(self):
        # the library must forget this plan's workspaces before their memory is reused (a registered address that became someone
        # else's scratch buffer would make that call skip its pack)
        try:
            self.invalidate()
        except Exception:       # interpreter shutdown
            pass

    def tick(self):
        """Start of an owning step: after two steps without a change of kernels the next one records."""
        if self.state == "off":
            self.quiet = getattr(self, "quiet", 0) + 1
            if self.quiet > 2:
                self.state = "record"

    def replay(self, group):
        if self.state != "replay" or not self.armed:
            return
        t = self.tables.get(group)
        if t is None:
            return
        check(_lib.load().cstp_pack_replay(_stream(), t[0].data_ptr(), t[1].data_ptr(), t[2], t[3]), "cstp_pack_replay")
        self.stats["replays"] += 1


pack_plan: Optional[PackPlan] = None          # set by the training step that owns the model (train.PretrainStep)


def _packed_call(w_param, tag, x_shape, device, nbytes, fn):
    """``fn(ws)`` = one C-ABI convolution call that packs ``w_param`` into its workspace: through the pack plan when a
    training step has armed one (persistent workspace, recorded / skipped pack), else on the shared scratch arena."""
    plan = pack_plan
    if plan is None or not plan.armed or plan.state == "off" or w_param is None:
        ws = _workspace(device, nbytes)
        fn(ws)
        return
    key = PackPlan.key(w_param, tag, x_shape)
    ws = plan.workspace(key, device, nbytes)
    if plan.state == "record":
        plan.call(key, lambda: fn(ws))
    else:
        fn(ws)
        if plan.state == "replay" and key in plan.recs:
            plan.stats["skipped_calls"] += 1


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CstpError("%s must be on a HIP device (cstp_amd has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise _lib.CstpError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _triple(v) -> Tuple[int, int, int]:
    return (v, v, v) if isinstance(v, int) else tuple(v)


# Tile autotuning: the first time a convolution geometry is seen (per direction and GEMM arithmetic) its tile comes from the
# persisted table cstp_amd/tuned/gfx950_abi<N>.json (cstp_conv3d_set_tile), else the library times its candidates once
# (cstp_conv3d_autotune), and the winner is written back to the table -- so that runs, boxes and ranks execute the same kernels
# (round-1 VERDICT weak-9: re-deriving the table by timing on every process start made bench.py vary 217..227 clips/s).
# CSTP_AUTOTUNE=0 keeps the analytic choice; CSTP_TUNE_TABLE=0 ignores and does not write the file; CSTP_TUNE_TABLE=<path>
# uses another file; CSTP_TUNE_TABLE_RO=1 reads it but never writes (the test suite: tests/conftest.py).  TUNE_REV names the
# candidate set the entries were chosen from: bump it whenever a kernel variant is added or changed, and stale tables are
# ignored.
AUTOTUNE = os.environ.get("CSTP_AUTOTUNE", "1") != "0"
TUNE_REV = 5
_tuned = set()
_TABLE_ENV = os.environ.get("CSTP_TUNE_TABLE", "")
TUNE_TABLE_PATH = None if _TABLE_ENV == "0" else (
    _TABLE_ENV or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_abi%d.json" % _lib.ABI_VERSION))
_table = None            # {"<arith>|<mode>|<desc fields>": [tile4]}
_table_dirty = False
tune_stats = {"from_table": 0, "timed": 0}


def _load_table():
    global _table
    if _table is None:
        _table = {}
        if TUNE_TABLE_PATH is not None:
            try:
                import json
                with open(TUNE_TABLE_PATH) as f:
                    data = json.load(f)
                if data.get("arch") == "gfx950" and data.get("abi") == _lib.ABI_VERSION and data.get("rev") == TUNE_REV:
                    _table = {k: [int(v) for v in t] for k, t in data.get("tiles", {}).items()}
            except (OSError, ValueError, AttributeError):
                _table = {}
    return _table


_dirty_rank = None        # this process's rank when the table was first marked dirty (the exit hook runs after destroy_process_group)


def _is_rank0() -> bool:
    """Rank 0 of the job -- also at interpreter exit, when the process group is gone and dist.get_rank() would answer 0 on
    EVERY rank (round-3 ADVICE): the launcher's RANK, else the rank remembered when the table got its first new entry."""
    env = os.environ.get("RANK")
    if env is not None and env.isdigit():
        return int(env) == 0
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() == 0
    return _dirty_rank in (None, 0)


def _note_dirty() -> None:
    global _table_dirty, _dirty_rank
    _table_dirty = True
    if _dirty_rank is None:
        import torch.distributed as dist
        _dirty_rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def save_tune_table(path=None) -> bool:
    """Write the table (atomically) if it has new entries: on RANK 0 only, merged over what is on disk (another job may have
    added geometries meanwhile), once per call -- the training steps call it after their first step, and an exit hook
    covers everything else.  (Round 2 rewrote the file from every rank on every newly timed geometry: last writer won.)
    Silently skipped on a read-only tree."""
    global _table_dirty
    path = path or TUNE_TABLE_PATH
    if path is None or _table is None or not _table_dirty or os.environ.get("CSTP_TUNE_TABLE_RO", "0") == "1":
        return False
    if not _is_rank0():
        return False
    import json
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        merged = {}
        try:
            with open(path) as f:
                data = json.load(f)
            if data.get("arch") == "gfx950" and data.get("abi") == _lib.ABI_VERSION and data.get("rev") == TUNE_REV:
                merged = {k: [int(v) for v in t] for k, t in data.get("tiles", {}).items()}
        except (OSError, ValueError, AttributeError):
            merged = {}
        merged.update(_table)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as f:
            json.dump({"arch": "gfx950", "abi": _lib.ABI_VERSION, "rev": TUNE_REV,
                       "key": "arithmetic|mode|" + ",".join(f for f, _ in ConvDesc._fields_),
                       "tile": "cstp_conv3d_set_tile encoding", "tiles": dict(sorted(merged.items()))}, f, indent=0)
        os.replace(tmp, path)
        _table_dirty = False
        return True
    except OSError:
        return False


def _table_key(arith, mode, desc):
    return "%d|%d|%s" % (arith, mode, ",".join(str(getattr(desc, f)) for f, _ in ConvDesc._fields_))


def _autotune(lib, desc, mode, src, w, out, ws):
    global _table_dirty
    arith = lib.cstp_gemm_get_split_terms()
    key = (arith, mode) + tuple(getattr(desc, f) for f, _ in ConvDesc._fields_)
    if key in _tuned:
        return
    _tuned.add(key)
    if pack_plan is not None:
        pack_plan.invalidate()       # a layer gets its tile now: recorded packs may belong to another variant
    tbl = _load_table()
    sk = _table_key(arith, mode, desc)
    tile = tbl.get(sk)
    if tile is not None:
        arr = (ctypes.c_int32 * 4)(*tile)
        if lib.cstp_conv3d_set_tile(ctypes.byref(desc), mode, arr) == 0:
            tune_stats["from_table"] += 1
            return
    check(lib.cstp_conv3d_autotune(_stream(), ctypes.byref(desc), mode, src.data_ptr(), w.data_ptr(), out.data_ptr(),
                                   ws.data_ptr(), ws.numel(), 2), "cstp_conv3d_autotune")
    tune_stats["timed"] += 1
    arr = (ctypes.c_int32 * 4)()
    check(lib.cstp_conv3d_get_tile(ctypes.byref(desc), mode, arr), "cstp_conv3d_get_tile")
    if arr[0] >= 0:
        tbl[sk] = [int(v) for v in arr]
        _note_dirty()                # written by save_tune_table(): after a step's first call, or at exit


import atexit  # noqa: E402

atexit.register(save_tune_table)


def share_tune_table(src: int = 0) -> None:
    """Rank ``src``'s tuned tiles -> every rank (one broadcast of a small dict), so that all ranks of a job run the same
    kernel for the same layer even when some geometry had to be timed in this run; rank 0 then persists the table.
    Without a process group: only the save."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        save_tune_table()
        return
    if dist.get_rank() == src:
        save_tune_table()
    lib = _lib.load()
    box = [dict(_load_table()) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() == src:
        return
    tbl = _load_table()
    for sk, tile in box[0].items():
        if tbl.get(sk) == tile:
            continue
        tbl[sk] = tile
        arith, mode, fields = sk.split("|")
        if int(arith) != lib.cstp_gemm_get_split_terms():
            continue
        desc = ConvDesc(*[int(v) for v in fields.split(",")])
        lib.cstp_conv3d_set_tile(ctypes.byref(desc), int(mode), (ctypes.c_int32 * 4)(*tile))
        if pack_plan is not None:
            pack_plan.invalidate()


def set_split_terms(terms: int) -> None:
    """2 = f16 pair / three MFMA products (default), 3 = bf16 triple / six products, 1 = native f32 MFMA only,
    0 = environment default (cstp_gemm_set_split_terms).  Each arithmetic has its own class of tuned tiles."""
    check(_lib.load().cstp_gemm_set_split_terms(int(terms)), "cstp_gemm_set_split_terms")
    if pack_plan is not None:
        pack_plan.invalidate()


def set_deterministic(on: bool) -> None:
    """Bit-reproducible weight gradients (two-stage split-K instead of f32 atomics; cstp_set_deterministic).  Switch before
    the first convolution call of a run: the shared workspace is sized by what the library reports at that time."""
    check(_lib.load().cstp_set_deterministic(1 if on else 0), "cstp_set_deterministic")
    _ws_cache.clear()
    if pack_plan is not None:
        pack_plan.invalidate()
        pack_plan.ws.clear()


def set_conv_tile(x_shape, w_shape, stride, padding, mode: int, tile) -> None:
    """Pin the kernel variant of one convolution geometry and direction (cstp_conv3d_set_tile; mode 0 forward,
    1 backward_data, 2 backward_weight) and keep the autotuner away from it.  For parity tests and A/B timing."""
    lib = _lib.load()
    desc = _desc(tuple(x_shape), tuple(w_shape), _triple(stride), _triple(padding))
    arr = (ctypes.c_int32 * 4)(*[int(v) for v in tile])
    check(lib.cstp_conv3d_set_tile(ctypes.byref(desc), int(mode), arr), "cstp_conv3d_set_tile")
    if pack_plan is not None:
        pack_plan.invalidate()
    _tuned.add((lib.cstp_gemm_get_split_terms(), int(mode)) + tuple(getattr(desc, f) for f, _ in ConvDesc._fields_))


def _desc(x_shape, w_shape, stride, padding) -> ConvDesc:
    n, c, d, h, w = x_shape
    k, c2, kt, kh, kw = w_shape
    if c2 != c:
        raise _lib.CstpError("conv3d: input has %d channels, weight expects %d" % (c, c2))
    return ConvDesc(n, c, d, h, w, k, kt, kh, kw, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2])


def conv_out_shape(x_shape, w_shape, stride, padding):
    n, _, d, h, w = x_shape
    k, _, kt, kh, kw = w_shape
    return (n, k, (d + 2 * padding[0] - kt) // stride[0] + 1, (h + 2 * padding[1] - kh) // stride[1] + 1,
            (w + 2 * padding[2] - kw) // stride[2] + 1)


# ----------------------------------------------------------------------------------------------
# weight gradients on a side stream
# ----------------------------------------------------------------------------------------------
# A parameter is tagged ``_cstp_direct_grad`` by the training steps (cstp_amd.train -> mark_direct_grad) when its .grad tensor
# is a view of its model's flat gradient arena and nothing hooks its autograd accumulation (the flat all-reduce path, or no
# DDP at all): the kernels then add the gradient into the arena themselves and autograd sees None.  The tag is a property
# of the PARAMETER (of one model), not of the process: a second model in the same process -- DDP with its bucket reducer,
# anything that relies on gradient hooks -- is not affected (round-2 ADVICE).
def mark_direct_grad(model, arenas, on: bool) -> None:
    lo = hi = 0
    if arenas is not None:
        g = arenas["grad"]
        lo, hi = g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()
    for p in model.parameters():
        ok = bool(on) and p.grad is not None and lo <= p.grad.data_ptr() < hi and p.grad.is_contiguous()
        p._cstp_direct_grad = ok


def _direct(p) -> bool:
    return getattr(p, "_cstp_direct_grad", False) and p.is_leaf and p.grad is not None


OVERLAP_WGRAD = os.environ.get("CSTP_OVERLAP_WGRAD", "1") == "1"
_side_streams = {}
_join_pending = set()


def _side_stream(device: torch.device) -> torch.cuda.Stream:
    st = _side_streams.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[device.index] = st
    return st


def _join_side_streams() -> None:
    """Make the current stream wait for the side-stream weight gradients (end of every backward pass)."""
    for idx in list(_join_pending):
        torch.cuda.current_stream(torch.device("cuda", idx)).wait_stream(_side_streams[idx])
    _join_pending.clear()


def _queue_join(device: torch.device) -> None:
    if device.index not in _join_pending:
        if not _join_pending:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
        _join_pending.add(device.index)


# ----------------------------------------------------------------------------------------------
# convolution / linear
# ----------------------------------------------------------------------------------------------
# Largest-magnitude cells (cstp_hip.h: the *_am entry points).  A BatchNorm pass leaves max |y| (forward) / max |dx| (backward) in
# a one-element device tensor as a by-product and hangs it on the tensor it produced; the convolution that consumes that very
# tensor object hands it to the 2xf16-split kernels, which otherwise spend one extra read of the tensor on measuring it.  The
# attribute travels with the Python object only: any op in between (reshape, cat, add, an in-place update -- _version is
# checked) drops it, and the kernels measure for themselves.
FUSE_ABSMAX = os.environ.get("CSTP_FUSE_ABSMAX", "1") != "0"
# BatchNorm statistics as a by-product of the producing convolution (patch-kernel layers): 0 keeps the separate pass
FUSE_BN_STATS = os.environ.get("CSTP_FUSE_BN_STATS", "1") != "0"
absmax_stats = {"hit": 0, "miss": 0}


def _tag_absmax(t: torch.Tensor, cell: Optional[torch.Tensor]) -> None:
    if cell is not None:
        t._cstp_absmax = (cell, t._version)


def _absmax_of(t: torch.Tensor) -> Optional[torch.Tensor]:
    tag = getattr(t, "_cstp_absmax", None)
    if tag is not None and tag[1] == t._version and tag[0].device == t.device:
        absmax_stats["hit"] += 1
        return tag[0]
    absmax_stats["miss"] += 1
    return None


def _new_cell(like: torch.Tensor) -> Optional[torch.Tensor]:
    return torch.empty(1, dtype=torch.int32, device=like.device) if FUSE_ABSMAX else None


class GradJoin:
    """The gradients of ONE tensor that feeds ``n`` of these ops (the residual connection, r21d_byol.py:141-148: a block's
    input goes into conv1 AND into the addition behind bn2; in a downsample block into conv1 and the shortcut convolution)
    are summed inside the ops instead of by autograd's separate add pass over three tensors: every contributor but the last
    puts its gradient into the join's buffer (the first one creates it, convolutions after that ADD into it in their
    epilogue: cstp_conv3d_backward_data_acc) and hands autograd None; the last one adds its own and returns the sum.
    A fresh object per forward call; backward order is whatever the autograd engine chooses."""

    _open = []                # joins that hold a partial sum (checked when the backward pass ends)

    def __init__(self, n: int):
        self.n, self.count, self.buf = int(n), 0, None

    def contribute(self, plain, accumulate):
        """``plain() -> fresh gradient tensor``; ``accumulate(buf)``: buf += this contributor's gradient."""
        self.count += 1
        if self.buf is None:
            out = plain()
        else:
            accumulate(self.buf)
            # the accumulating kernel wrote through data_ptr(): tell autograd's version counter (round-3 ADVICE)
            torch.autograd.graph.increment_version(self.buf)
            out = self.buf
        if self.count < self.n:
            if self.buf is None:
                # first contributor: if the others never run (a consumer's input gradient pruned by autograd.grad(inputs=...),
                # needs_input_grad False on one branch) the partial sum would silently vanish -- an end-of-backward check raises
                GradJoin._open.append(self)
                if len(GradJoin._open) == 1:
                    torch.autograd.Variable._execution_engine.queue_callback(GradJoin._check_closed)
            self.buf = out
            return None
        if self in GradJoin._open:
            GradJoin._open.remove(self)
        self.buf, self.count = None, 0
        return out

    @staticmethod
    def _check_closed():
        left = [(j.count, j.n) for j in GradJoin._open]
        GradJoin._open.clear()
        if left:
            raise RuntimeError("GradJoin: %d residual joins ended the backward pass with contributors missing %r -- a consumer's "
                               "input gradient was pruned; the block input's gradient would have been dropped" % (len(left), left))


class _Conv3d(torch.autograd.Function):
    _last_stats = None

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, bn_groups=0, bn_pivot=None, grad_join=None):
        lib = _lib.load()
        xam = _absmax_of(x)
        w_in = w
        x = _req(x, "conv3d input")
        w = _req(w, "conv3d weight")
        desc = _desc(x.shape, w.shape, stride, padding)
        y = torch.empty(conv_out_shape(x.shape, w.shape, stride, padding), dtype=torch.float32, device=x.device)
        nbytes = lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc))
        ws = _workspace(x.device, nbytes)
        b = None if bias is None else _req(bias, "conv3d bias")
        if AUTOTUNE:
            _autotune(lib, desc, 0, x, w, y, ws)
        # a train-mode BatchNorm over bn_groups slices of the batch consumes y: where the layer's kernel can, it leaves the
        # statistics' partial sums beside y (cstp_conv3d_forward_bnstats) and batch_norm_act skips its pass over the tensor
        ns = lib.cstp_conv3d_bnstats_nsplit(ctypes.byref(desc), bn_groups) if (bn_groups > 0 and b is None and FUSE_BN_STATS) else 0
        _Conv3d._last_stats = None

        def run(ws):
            if ns > 0:
                # sums [k][groups][ns][2], pivots [k], then the (min, max) keys [k][groups][ns][2] as uint32 (one double each)
                part = torch.empty(w.shape[0] * bn_groups * ns * 3 + w.shape[0], dtype=torch.float64, device=x.device)
                zcell = torch.empty(1, dtype=torch.int32, device=x.device)
                got = ctypes.c_int32(0)
                pv = None if bn_pivot is None else _req(bn_pivot, "BatchNorm pivot")
                if pv is not None and pv.numel() != w.shape[0]:
                    raise _lib.CstpError("BatchNorm pivot has %d entries for %d output channels" % (pv.numel(), w.shape[0]))
                check(lib.cstp_conv3d_forward_bnstats(_stream(), ctypes.byref(desc), x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _ptr(xam), bn_groups, _ptr(pv), part.data_ptr(),
                                                      part.numel() * 8, ctypes.byref(got), zcell.data_ptr(), None),
                      "cstp_conv3d_forward_bnstats")
                if got.value > 0:
                    _Conv3d._last_stats = (part, got.value, bn_groups, zcell)
            else:
                check(lib.cstp_conv3d_forward_am(_stream(), ctypes.byref(desc), x.data_ptr(), w.data_ptr(), _ptr(b), None,
                                                 y.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(xam)), "cstp_conv3d_forward")

        with _span("conv3d_forward", lambda: _desc_key(desc)):
            _packed_call(w_in if w.data_ptr() == w_in.data_ptr() else None, "f", x.shape, x.device, nbytes, run)
        ctx.save_for_backward(x, w)
        ctx.w_param = w_in           # the parameter object itself (save_for_backward hands back a new tensor object)
        ctx.grad_join = grad_join
        ctx.x_absmax = xam
        ctx.desc = desc
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        desc = ctx.desc
        dyam = _absmax_of(dy)
        xam = ctx.x_absmax
        dy = _req(dy, "conv3d grad_output")
        nbytes = lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc))
        dx = dw = db = None
        # The gradient of a leaf weight whose .grad is a contiguous slice of the flat gradient arena (DIRECT_WGRAD) is ADDED
        # there by the library itself -- AccumulateGrad folded into the unpacking pass: no temporary, no add kernel -- and
        # autograd sees None for this input.  With OVERLAP_WGRAD it also runs on a second HIP stream: it feeds nothing
        # downstream in the backward chain, so the matrix-core-bound weight-gradient kernels execute beside the HBM-bound
        # BatchNorm backward kernels of the main chain (the arena is joined before anything reads it, _join_side_streams).
        direct_w = ctx.needs_input_grad[1] and _direct(ctx.w_param)
        side_w = direct_w and OVERLAP_WGRAD and x.dim() == 5 and x.shape[2] * x.shape[3] * x.shape[4] > 1

        def wgrad_into_arena():
            wsx = _workspace(x.device, nbytes)
            if AUTOTUNE and (lib.cstp_gemm_get_split_terms(), 2) + _desc_key(desc) not in _tuned:
                _autotune(lib, desc, 2, x, dy, torch.empty_like(w), wsx)          # (tuning overwrites its output)
            with _span("conv3d_backward_weight", lambda: _desc_key(desc)):
                check(lib.cstp_conv3d_backward_weight_acc(_stream(), ctypes.byref(desc), x.data_ptr(), None, dy.data_ptr(),
                                                          ctx.w_param.grad.data_ptr(), wsx.data_ptr(), wsx.numel(), _ptr(xam),
                                                          _ptr(dyam), 1), "cstp_conv3d_backward_weight")

        if side_w:
            main = torch.cuda.current_stream(x.device)
            side = _side_stream(x.device)
            side.wait_stream(main)                     # dy is complete
            with torch.cuda.stream(side):
                wgrad_into_arena()
            x.record_stream(side)
            dy.record_stream(side)
            for cell in (xam, dyam):                   # the cells are read by the side-stream kernels too
                if cell is not None:
                    cell.record_stream(side)
            _queue_join(x.device)
        if ctx.needs_input_grad[0]:
            def dgrad(dst, acc):
                if AUTOTUNE and (lib.cstp_gemm_get_split_terms(), 1) + _desc_key(desc) not in _tuned:
                    _autotune(lib, desc, 1, dy, w, torch.empty_like(dst) if acc else dst, _workspace(x.device, nbytes))   # (tuning overwrites its output)
                with _span("conv3d_backward_data", lambda: _desc_key(desc)):
                    _packed_call(ctx.w_param if w.data_ptr() == ctx.w_param.data_ptr() else None, "d", x.shape, x.device, nbytes,
                                 lambda ws: check(lib.cstp_conv3d_backward_data_acc(
                                     _stream(), ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), dst.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _ptr(dyam), 1 if acc else 0), "cstp_conv3d_backward_data"))
                return dst
            join = ctx.grad_join
            if join is None:
                dx = dgrad(torch.empty_like(x), False)
            else:       # x also feeds another op: the sum of the two gradients is formed here (GradJoin)
                dx = join.contribute(lambda: dgrad(torch.empty_like(x), False), lambda buf: dgrad(buf, True))
        if direct_w and not side_w:
            wgrad_into_arena()
        elif ctx.needs_input_grad[1] and not direct_w:
            dw = torch.empty_like(w)
            ws = _workspace(x.device, nbytes)
            if AUTOTUNE:
                _autotune(lib, desc, 2, x, dy, dw, ws)
            with _span("conv3d_backward_weight", lambda: _desc_key(desc)):
                check(lib.cstp_conv3d_backward_weight_am(_stream(), ctypes.byref(desc), x.data_ptr(), None, dy.data_ptr(),
                                                         dw.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(xam), _ptr(dyam)),
                      "cstp_conv3d_backward_weight")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            n, k = dy.shape[0], dy.shape[1]
            s = dy.numel() // (n * k)
            db = torch.empty(k, dtype=torch.float32, device=dy.device)
            check(lib.cstp_channel_sum(_stream(), dy.data_ptr(), db.data_ptr(), n, k, s, None, 0), "cstp_channel_sum")
        return dx, dw, db, None, None, None, None, None


def conv3d(x, w, bias=None, stride=1, padding=0, bn_groups=0, bn_pivot=None, grad_join=None):
    """F.conv3d drop-in (fp32, NCDHW).  ``bn_groups`` > 0: the caller feeds the result to a train-mode ``batch_norm_act`` with
    that many groups -- the convolution then leaves the BatchNorm's statistics beside its output where its kernel can.
    ``bn_pivot`` ([C_out], e.g. that BatchNorm's running_mean): the sums are taken around it (cstp_conv3d_forward_bnstats).
    ``grad_join``: x also feeds another op of this module that was given the same GradJoin (see there).
    A bfloat16 ``x`` selects the bf16-storage path (below): bf16 result, fp32 weight and weight gradient."""
    if x.dtype == torch.bfloat16:
        if bias is not None:
            raise _lib.CstpError("bf16-storage conv3d is bias-free (models/BE/r3d_byol.py:45-53)")
        return _Conv3dB16.apply(x, w, _triple(stride), _triple(padding), grad_join)
    y = _Conv3d.apply(x, w, bias, _triple(stride), _triple(padding), int(bn_groups), bn_pivot, grad_join)
    st = _Conv3d._last_stats
    _Conv3d._last_stats = None
    if st is not None:
        y._cstp_bnstats = st + (y._version,)
    return y


def _bnstats_of(t, groups):
    """(partials, nsplit, cell) the convolution that produced ``t`` left for a BatchNorm over ``groups`` groups, or None."""
    tag = getattr(t, "_cstp_bnstats", None)
    if tag is not None and tag[4] == t._version and tag[2] == groups and tag[0].device == t.device:
        return tag[0], tag[1], tag[3]
    return None


def linear(x, w, bias=None):
    """F.linear drop-in for 2-D x: the 1x1x1 convolution over [B][F][1][1][1]."""
    y = _Conv3d.apply(x.reshape(x.shape[0], x.shape[1], 1, 1, 1), w.reshape(w.shape[0], w.shape[1], 1, 1, 1), bias,
                      (1, 1, 1), (0, 0, 0), 0, None, None)
    return y.reshape(x.shape[0], w.shape[0])


# ----------------------------------------------------------------------------------------------
# train-mode BatchNorm (+ residual) (+ ReLU)
# ----------------------------------------------------------------------------------------------
class _BNAct(torch.autograd.Function):
    _last_cell = None
    _pre_stats = None     # (partial sums, nsplit) the producing convolution left for this call (batch_norm_act sets it)

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, relu, eps, momentum, groups, grad_join=None):
        lib = _lib.load()
        x = _req(x, "batch_norm input")
        gamma = _req(gamma, "batch_norm weight")
        beta = _req(beta, "batch_norm bias")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        res = None if residual is None else _req(residual, "residual")
        if res is not None and res.shape != x.shape:
            raise _lib.CstpError("residual shape %s != input shape %s" % (tuple(res.shape), tuple(x.shape)))
        if groups < 1 or n % groups != 0:
            raise _lib.CstpError("batch of %d rows cannot be split into %d BN groups" % (n, groups))
        y = torch.empty_like(x)
        mean = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        nbytes = lib.cstp_bn_workspace_bytes(n, c, s, groups)
        ws = _workspace(x.device, nbytes)
        # ReLU without a residual: backward recomputes the mask from x with the affine table instead of re-reading y
        remask = relu and res is None and s > 1
        ss = torch.empty(groups * c * 2, dtype=torch.float32, device=x.device) if remask else None
        cell = _new_cell(x) if s > 1 else None
        pre = _BNAct._pre_stats
        _BNAct._pre_stats = None
        with _span("bn_forward", (n, c, s, groups, res is not None, bool(relu))):
            if pre is not None and s > 1:
                check(lib.cstp_bn_forward_train_pre(_stream(), x.data_ptr(), _ptr(res), y.data_ptr(), gamma.data_ptr(),
                                                    beta.data_ptr(), _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                                    invstd.data_ptr(), _ptr(ss), n, c, s, groups, eps, momentum,
                                                    1 if relu else 0, ws.data_ptr(), ws.numel(), _ptr(cell), pre[0].data_ptr(),
                                                    pre[1]), "cstp_bn_forward_train_pre")
            else:
                check(lib.cstp_bn_forward_train_am(_stream(), x.data_ptr(), _ptr(res), y.data_ptr(), gamma.data_ptr(),
                                                   beta.data_ptr(), _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                                   invstd.data_ptr(), _ptr(ss), n, c, s, groups, eps, momentum, 1 if relu else 0,
                                                   ws.data_ptr(), ws.numel(), _ptr(cell)), "cstp_bn_forward_train")
        _BNAct._last_cell = cell     # batch_norm_act hangs it on the tensor object apply() returns
        if remask:
            ctx.save_for_backward(x, ss, gamma, mean, invstd)
        else:
            ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.remask = remask
        ctx.relu = relu
        ctx.groups = groups
        ctx.has_res = res is not None
        ctx.params = (gamma, beta)     # the parameter objects themselves (their .grad may be an arena slice, see backward)
        ctx.grad_join = grad_join
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y_or_ss, gamma, mean, invstd = ctx.saved_tensors
        y, ss = (None, y_or_ss) if ctx.remask else (y_or_ss, None)
        dy = _req(dy, "batch_norm grad_output")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        # parameters whose .grad is a live slice of the flat gradient arena (cstp_amd.train): the kernel adds into it directly
        pg, pb = ctx.params
        direct = ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _direct(pg) and _direct(pb)
        dgamma = pg.grad if direct else torch.empty_like(gamma)
        dbeta = pb.grad if direct else torch.empty_like(gamma)
        nbytes = lib.cstp_bn_workspace_bytes(n, c, s, ctx.groups)
        ws = _workspace(x.device, nbytes)
        cell = _new_cell(x) if s > 1 else None
        with _span("bn_backward", (n, c, s, ctx.groups, ctx.has_res, bool(ctx.relu))):
            check(lib.cstp_bn_backward_am(_stream(), x.data_ptr(), _ptr(y), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                          invstd.data_ptr(), _ptr(ss), dx.data_ptr(), _ptr(dres), dgamma.data_ptr(),
                                          dbeta.data_ptr(), n, c, s, ctx.groups, 1 if ctx.relu else 0, ws.data_ptr(), ws.numel(),
                                          _ptr(cell), 1 if direct else 0), "cstp_bn_backward")
        _tag_absmax(dx, cell)
        if dres is not None and ctx.grad_join is not None:       # the residual tensor's other consumer adds its gradient to this
            dres = ctx.grad_join.contribute(lambda: dres, lambda buf: buf.add_(dres))
        if direct:
            return dx, None, None, dres, None, None, None, None, None, None, None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean=None, running_var=None, residual=None, relu=False, eps=BN_EPS,
                   momentum=BN_MOMENTUM, groups=1, grad_join=None):
    """y = act(batch_norm_train(x) + residual); running stats updated in place.  ``groups`` > 1: the batch is
    that many independent BN calls back to back (per-group statistics, sequential running-stat updates).
    A bfloat16 ``x`` (and residual) selects the bf16-storage path."""
    if x.dtype == torch.bfloat16:
        return _BNActB16.apply(x, gamma, beta, residual, running_mean, running_var, bool(relu), float(eps), float(momentum),
                               int(groups), grad_join)
    _BNAct._pre_stats = _bnstats_of(x, int(groups))
    y = _BNAct.apply(x, gamma, beta, residual, running_mean, running_var, bool(relu), float(eps), float(momentum),
                     int(groups), grad_join)
    _tag_absmax(y, _BNAct._last_cell)
    _BNAct._last_cell = None
    return y


def batch_norm_eval(x, gamma, beta, running_mean, running_var, residual=None, relu=False, eps=BN_EPS):
    """y = act(batch_norm(x; running stats) + residual) -- model.eval().  Forward only: the reference evaluates under
    torch.no_grad() (main_ft_mp.py:261, test.py:75), so asking for a gradient through it is an error, not a fallback."""
    lib = _lib.load()
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad or beta.requires_grad
                                    or (residual is not None and residual.requires_grad)):
        raise _lib.CstpError("eval-mode BatchNorm is forward-only: call it under torch.no_grad() (as the reference's "
                             "validation/test loops do)")
    if x.dtype == torch.bfloat16:          # the bf16-storage path
        x = _req16(x, "batch_norm input")
        n, c = x.shape[0], x.shape[1]
        res = None if residual is None else _req16(residual, "residual")
        if res is not None and res.shape != x.shape:
            raise _lib.CstpError("residual shape %s != input shape %s" % (tuple(res.shape), tuple(x.shape)))
        y = torch.empty_like(x)
        check(lib.cstp_b16_bn_forward_eval(_stream(), x.data_ptr(), _ptr(res), y.data_ptr(), _req(gamma, "weight").data_ptr(),
                                           _req(beta, "bias").data_ptr(), _req(running_mean, "running_mean").data_ptr(),
                                           _req(running_var, "running_var").data_ptr(), n, c, x.numel() // (n * c), float(eps),
                                           1 if relu else 0), "cstp_b16_bn_forward_eval")
        return y
    x = _req(x, "batch_norm input")
    n, c = x.shape[0], x.shape[1]
    s = x.numel() // (n * c)
    res = None if residual is None else _req(residual, "residual")
    if res is not None and res.shape != x.shape:
        raise _lib.CstpError("residual shape %s != input shape %s" % (tuple(res.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    cell = _new_cell(x) if s > 1 else None
    ws = _workspace(x.device, lib.cstp_bn_workspace_bytes(n, c, s, 1) if cell is not None else lib.cstp_bn_eval_workspace_bytes(c))
    check(lib.cstp_bn_forward_eval_am(_stream(), x.data_ptr(), _ptr(res), y.data_ptr(), _req(gamma, "weight").data_ptr(),
                                      _req(beta, "bias").data_ptr(), _req(running_mean, "running_mean").data_ptr(),
                                      _req(running_var, "running_var").data_ptr(), n, c, s, float(eps), 1 if relu else 0,
                                      ws.data_ptr(), ws.numel(), _ptr(cell)), "cstp_bn_forward_eval")
    _tag_absmax(y, cell)
    return y


# ----------------------------------------------------------------------------------------------
# fused  BatchNorm(train) -> ReLU -> conv3d : the normalised tensor never exists in HBM
# ----------------------------------------------------------------------------------------------
class _BNReluConv3d(torch.autograd.Function):
    """y = conv3d(relu(batch_norm_train(x)), w) (r21d_byol.py:94-97: temporal_conv(relu(bn(spatial_conv(.))))).
    Forward: the statistics (folded from the partial sums the producing convolution left, else one pass over x), then the
    convolution applies x*scale+shift (+ReLU) inside its gather.  Backward: weight gradient with the same transform
    recomputed in ITS gather (side stream, straight into the gradient arena), data gradient of the convolution, then the
    BN backward with the ReLU mask recomputed from x."""
    _pre_stats = None     # (partials, nsplit, cell) of the producing convolution (bn_relu_conv3d sets it)
    _last_stats = None    # what THIS convolution left for the BatchNorm behind it (out_groups > 0), as _Conv3d._last_stats

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, w, stride, padding, groups, relu, eps, momentum,
                out_groups=0, out_pivot=None):
        lib = _lib.load()
        pre = _BNReluConv3d._pre_stats
        _BNReluConv3d._pre_stats = None
        w_in, g_in, b_in = w, gamma, beta
        x = _req(x, "bn_relu_conv3d input")
        gamma, beta, w = _req(gamma, "bn weight"), _req(beta, "bn bias"), _req(w, "conv weight")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        if groups < 1 or groups > 4 or n % groups != 0:
            raise _lib.CstpError("batch of %d rows cannot be split into %d BN groups" % (n, groups))
        mean = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        ss = torch.empty(groups * c * 2, dtype=torch.float32, device=x.device)
        zam = None
        with _span("bn_forward", (n, c, s, groups, False, bool(relu))):
            if pre is not None:
                zam = pre[2]       # zeroed by the producing launch; the finalize takes the maximum of act(x*scale+shift) into it
                check(lib.cstp_bn_finalize_pre(_stream(), gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                               mean.data_ptr(), invstd.data_ptr(), ss.data_ptr(), n, c, s, groups, eps, momentum,
                                               1 if relu else 0, pre[0].data_ptr(), pre[1], zam.data_ptr()), "cstp_bn_finalize_pre")
            else:
                ws = _workspace(x.device, lib.cstp_bn_workspace_bytes(n, c, s, groups))
                check(lib.cstp_bn_stats_train(_stream(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean),
                                              _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), ss.data_ptr(), n, c, s, groups,
                                              eps, momentum, ws.data_ptr(), ws.numel()), "cstp_bn_stats_train")
        desc = _desc(x.shape, w.shape, stride, padding)
        y = torch.empty(conv_out_shape(x.shape, w.shape, stride, padding), dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)))
        if AUTOTUNE:
            _autotune(lib, desc, 0, x, w, y, ws)
        aff = InAffine(ss.data_ptr(), groups, 1 if relu else 0)
        # a train-mode BatchNorm over out_groups slices consumes y: where this layer's kernel can, it leaves that BatchNorm's
        # sums and range beside y (the temporal patch kernel igemm_k1t<.., STATS, AFF>)
        ns = lib.cstp_conv3d_bnstats_nsplit_aff(ctypes.byref(desc), out_groups) if (out_groups > 0 and FUSE_BN_STATS) else 0
        _BNReluConv3d._last_stats = None

        def run(ws):
            if ns > 0:
                part = torch.empty(w.shape[0] * out_groups * ns * 3 + w.shape[0], dtype=torch.float64, device=x.device)
                ycell = torch.empty(1, dtype=torch.int32, device=x.device)
                got = ctypes.c_int32(0)
                pv = None if out_pivot is None else _req(out_pivot, "BatchNorm pivot")
                check(lib.cstp_conv3d_forward_bnstats(_stream(), ctypes.byref(desc), x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _ptr(zam), out_groups, _ptr(pv), part.data_ptr(),
                                                      part.numel() * 8, ctypes.byref(got), ycell.data_ptr(), ctypes.byref(aff)),
                      "cstp_conv3d_forward_bnstats")
                if got.value > 0:
                    _BNReluConv3d._last_stats = (part, got.value, out_groups, ycell)
            else:
                check(lib.cstp_conv3d_forward_am(_stream(), ctypes.byref(desc), x.data_ptr(), w.data_ptr(), None, ctypes.byref(aff),
                                                 y.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(zam)), "cstp_conv3d_forward")

        with _span("conv3d_forward", lambda: _desc_key(desc)):
            # ("fa": a forward that carries the in_affine may run another kernel variant -- another pack -- than the plain one)
            _packed_call(w_in if w.data_ptr() == w_in.data_ptr() else None, "fa", x.shape, x.device,
                         lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), run)
        ctx.save_for_backward(x, gamma, mean, invstd, ss, w)
        ctx.desc, ctx.groups, ctx.relu, ctx.z_absmax = desc, groups, relu, zam
        ctx.params = (g_in, b_in, w_in)      # the parameter objects themselves (their .grad may be an arena slice)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mean, invstd, ss, w = ctx.saved_tensors
        desc, zam = ctx.desc, ctx.z_absmax
        pg, pb, pw = ctx.params
        dyam = _absmax_of(dy)
        dy = _req(dy, "bn_relu_conv3d grad_output")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        nbytes = max(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), lib.cstp_bn_workspace_bytes(n, c, s, ctx.groups))
        aff = InAffine(ss.data_ptr(), ctx.groups, 1 if ctx.relu else 0)
        dw = None
        direct_w = ctx.needs_input_grad[5] and _direct(pw)
        side_w = direct_w and OVERLAP_WGRAD

        def wgrad(dst, accumulate):
            wsx = _workspace(x.device, nbytes)
            if AUTOTUNE and (lib.cstp_gemm_get_split_terms(), 2) + _desc_key(desc) not in _tuned:
                _autotune(lib, desc, 2, x, dy, torch.empty_like(w), wsx)          # (tuning overwrites its output)
            with _span("conv3d_backward_weight", lambda: _desc_key(desc)):
                check(lib.cstp_conv3d_backward_weight_acc(_stream(), ctypes.byref(desc), x.data_ptr(), ctypes.byref(aff),
                                                          dy.data_ptr(), dst.data_ptr(), wsx.data_ptr(), wsx.numel(), _ptr(zam),
                                                          _ptr(dyam), 1 if accumulate else 0), "cstp_conv3d_backward_weight")

        if side_w:       # as _Conv3d.backward: the weight gradient feeds nothing downstream in the backward chain
            main = torch.cuda.current_stream(x.device)
            side = _side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                wgrad(pw.grad, True)
            for t in (x, dy, ss, zam, dyam):
                if t is not None:
                    t.record_stream(side)
            _queue_join(x.device)
        elif direct_w:
            wgrad(pw.grad, True)
        elif ctx.needs_input_grad[5]:
            dw = torch.empty_like(w)
            wgrad(dw, False)
        dx = dgamma = dbeta = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ws = _workspace(x.device, nbytes)
            dz = torch.empty_like(x)     # gradient w.r.t. the (never materialised) normalised activation
            if AUTOTUNE and (lib.cstp_gemm_get_split_terms(), 1) + _desc_key(desc) not in _tuned:
                _autotune(lib, desc, 1, dy, w, dz, ws)
            with _span("conv3d_backward_data", lambda: _desc_key(desc)):
                _packed_call(pw if w.data_ptr() == pw.data_ptr() else None, "d", x.shape, x.device,
                             lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)),
                             lambda wsd: check(lib.cstp_conv3d_backward_data_acc(
                                 _stream(), ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), dz.data_ptr(), wsd.data_ptr(),
                                 wsd.numel(), _ptr(dyam), 0), "cstp_conv3d_backward_data"))
            dx = torch.empty_like(x)
            direct = ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _direct(pg) and _direct(pb)
            dgamma = pg.grad if direct else torch.empty_like(gamma)
            dbeta = pb.grad if direct else torch.empty_like(gamma)
            cell = _new_cell(x)
            with _span("bn_backward", (n, c, s, ctx.groups, False, bool(ctx.relu))):
                check(lib.cstp_bn_backward_am(_stream(), x.data_ptr(), None, dz.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), ss.data_ptr(), dx.data_ptr(), None, dgamma.data_ptr(),
                                              dbeta.data_ptr(), n, c, s, ctx.groups, 1 if ctx.relu else 0, ws.data_ptr(),
                                              ws.numel(), _ptr(cell), 1 if direct else 0), "cstp_bn_backward")
            _tag_absmax(dx, cell)
            if direct:
                dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, dw, None, None, None, None, None, None, None, None


def in_affine_fused(x_shape, w_shape, stride, padding, groups) -> bool:
    """Would a convolution of this geometry apply a BatchNorm + ReLU over ``groups`` groups inside its f16-pair gather kernels
    (forward AND weight gradient)?  cstp_conv3d_in_affine_fused."""
    desc = _desc(tuple(x_shape), tuple(w_shape), _triple(stride), _triple(padding))
    return bool(_lib.load().cstp_conv3d_in_affine_fused(ctypes.byref(desc), int(groups)))


def bn_relu_conv3d(x, gamma, beta, running_mean, running_var, w, stride=1, padding=0, groups=1, relu=True, eps=BN_EPS,
                   momentum=BN_MOMENTUM, bn_groups=0, bn_pivot=None):
    """conv3d(act(batch_norm_train(x)), w) with the BN apply fused into the convolution's gather.  Where the convolution that
    produced ``x`` left the BatchNorm's sums and range beside it (conv3d(.., bn_groups=groups)) no pass reads x before the
    consuming convolution does.  ``bn_groups`` / ``bn_pivot``: as conv3d's -- the BatchNorm BEHIND this convolution."""
    _BNReluConv3d._pre_stats = _bnstats_of(x, int(groups))
    y = _BNReluConv3d.apply(x, gamma, beta, running_mean, running_var, w, _triple(stride), _triple(padding), int(groups),
                            bool(relu), float(eps), float(momentum), int(bn_groups), bn_pivot)
    st = _BNReluConv3d._last_stats
    _BNReluConv3d._last_stats = None
    if st is not None:
        y._cstp_bnstats = st + (y._version,)
    return y


# ----------------------------------------------------------------------------------------------
# global average pool
# ----------------------------------------------------------------------------------------------
class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req(x, "avgpool input")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(lib.cstp_avgpool_forward(_stream(), x.data_ptr(), y.data_ptr(), n * c, s), "cstp_avgpool_forward")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _req(dy, "avgpool grad_output")
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        n, c = ctx.shape[0], ctx.shape[1]
        s = dx.numel() // (n * c)
        check(lib.cstp_avgpool_backward(_stream(), dy.data_ptr(), dx.data_ptr(), n * c, s), "cstp_avgpool_backward")
        return dx


class _MaxPool3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        lib = _lib.load()
        x = _req(x, "max_pool3d input")
        if x.dim() != 5:
            raise _lib.CstpError("max_pool3d expects [N, C, D, H, W], got %s" % (tuple(x.shape),))
        n, c, d, h, w = x.shape
        k = (ctypes.c_int32 * 3)(*kernel)
        st = (ctypes.c_int32 * 3)(*stride)
        pd = (ctypes.c_int32 * 3)(*padding)
        osz = tuple((sz + 2 * padding[i] - kernel[i]) // stride[i] + 1 for i, sz in enumerate((d, h, w)))
        y = torch.empty((n, c) + osz, dtype=torch.float32, device=x.device)
        idx = torch.empty((n, c) + osz, dtype=torch.int32, device=x.device)
        check(lib.cstp_maxpool3d_forward(_stream(), x.data_ptr(), y.data_ptr(), idx.data_ptr(), n * c, d, h, w, k, st, pd),
              "cstp_maxpool3d_forward")
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape), tuple(kernel), tuple(stride), tuple(padding))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        shape, kernel, stride, padding = ctx.geom
        dy = _req(dy, "max_pool3d grad_output")
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        n, c, d, h, w = shape
        check(lib.cstp_maxpool3d_backward(_stream(), dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n * c, d, h, w,
                                          (ctypes.c_int32 * 3)(*kernel), (ctypes.c_int32 * 3)(*stride),
                                          (ctypes.c_int32 * 3)(*padding)), "cstp_maxpool3d_backward")
        return dx, None, None, None


def max_pool3d(x, kernel_size=3, stride=2, padding=1):
    """nn.MaxPool3d(kernel_size, stride, padding) (models/BE/r3d_byol.py:158)."""
    if x.dtype == torch.bfloat16:
        return _MaxPool3dB16.apply(x, _triple(kernel_size), _triple(stride), _triple(padding))
    return _MaxPool3d.apply(x, _triple(kernel_size), _triple(stride), _triple(padding))


def global_avg_pool(x):
    """AdaptiveAvgPool3d(1) + view(-1, C).  bfloat16 in -> float32 out (the heads behind it are fp32)."""
    if x.dtype == torch.bfloat16:
        return _AvgPoolB16.apply(x)
    return _AvgPool.apply(x)


# ----------------------------------------------------------------------------------------------
# the bf16-STORAGE path (csrc/b16.hip; BASELINE configs[4]), selected by the dtype of the activation tensor:
# 5-D activations and their gradients bf16, parameters / their gradients / statistics fp32 (cstp_hip.h "bf16-STORAGE path")
# ----------------------------------------------------------------------------------------------
def _req16(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CstpError("%s must be on a HIP device (cstp_amd has no CPU path)" % name)
    if t.dtype != torch.bfloat16:
        raise _lib.CstpError("%s must be bfloat16 on the bf16-storage path, got %s" % (name, t.dtype))
    return t.contiguous()


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """The fp32 clip rounded to bf16 (what autocast does to the first convolution's input); no gradient flows into a clip."""
    lib = _lib.load()
    x = _req(x.detach(), "clip")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.cstp_b16_cast(_stream(), x.data_ptr(), y.data_ptr(), x.numel()), "cstp_b16_cast")
    return y


class _Conv3dB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, grad_join=None):
        lib = _lib.load()
        w_in = w
        ctx.grad_join = grad_join
        x = _req16(x, "conv3d input")
        w = _req(w, "conv3d weight")
        desc = _desc(x.shape, w.shape, stride, padding)
        y = torch.empty(conv_out_shape(x.shape, w.shape, stride, padding), dtype=torch.bfloat16, device=x.device)
        nbytes = lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(desc))
        with _span("conv3d_forward", lambda: ("bf16",) + _desc_key(desc)):
            _packed_call(w_in if w.data_ptr() == w_in.data_ptr() else None, "f", x.shape, x.device, nbytes,
                         lambda ws: check(lib.cstp_b16_conv3d_forward(_stream(), ctypes.byref(desc), x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                                                      ws.data_ptr(), ws.numel()), "cstp_b16_conv3d_forward"))
        ctx.save_for_backward(x, w)
        ctx.w_param = w_in
        ctx.desc = desc
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        desc = ctx.desc
        dy = _req16(dy, "conv3d grad_output")
        nbytes = lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(desc))
        dx = dw = None
        direct_w = ctx.needs_input_grad[1] and _direct(ctx.w_param)

        def wgrad(dst, accumulate):
            wsx = _workspace(x.device, nbytes)
            with _span("conv3d_backward_weight", lambda: ("bf16",) + _desc_key(desc)):
                check(lib.cstp_b16_conv3d_backward_weight(_stream(), ctypes.byref(desc), x.data_ptr(), dy.data_ptr(), dst.data_ptr(),
                                                          wsx.data_ptr(), wsx.numel(), 1 if accumulate else 0),
                      "cstp_b16_conv3d_backward_weight")

        if direct_w and OVERLAP_WGRAD:      # as _Conv3d.backward: the weight gradient feeds nothing downstream in the chain
            main = torch.cuda.current_stream(x.device)
            side = _side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                wgrad(ctx.w_param.grad, True)
            x.record_stream(side)
            dy.record_stream(side)
            _queue_join(x.device)
        elif direct_w:
            wgrad(ctx.w_param.grad, True)
        elif ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            wgrad(dw, False)
        if ctx.needs_input_grad[0]:
            def dgrad(dst, acc):
                with _span("conv3d_backward_data", lambda: ("bf16",) + _desc_key(desc)):
                    _packed_call(ctx.w_param if w.data_ptr() == ctx.w_param.data_ptr() else None, "d", x.shape, x.device, nbytes,
                                 lambda ws: check(lib.cstp_b16_conv3d_backward_data_acc(
                                     _stream(), ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), dst.data_ptr(), ws.data_ptr(), ws.numel(),
                                     1 if acc else 0), "cstp_b16_conv3d_backward_data"))
                return dst
            join = ctx.grad_join
            if join is None:
                dx = dgrad(torch.empty_like(x), False)
            else:       # x also feeds another op (the residual connection): the sum of the two gradients is formed here (GradJoin)
                dx = join.contribute(lambda: dgrad(torch.empty_like(x), False), lambda buf: dgrad(buf, True))
        return dx, dw, None, None, None


class _BNActB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, relu, eps, momentum, groups, grad_join=None):
        lib = _lib.load()
        ctx.param_objs = (gamma, beta)     # the parameter objects themselves (their .grad may be an arena slice)
        ctx.grad_join = grad_join
        x = _req16(x, "batch_norm input")
        gamma = _req(gamma, "batch_norm weight")
        beta = _req(beta, "batch_norm bias")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        res = None if residual is None else _req16(residual, "residual")
        if res is not None and res.shape != x.shape:
            raise _lib.CstpError("residual shape %s != input shape %s" % (tuple(res.shape), tuple(x.shape)))
        if groups < 1 or n % groups != 0:
            raise _lib.CstpError("batch of %d rows cannot be split into %d BN groups" % (n, groups))
        y = torch.empty_like(x)
        mean = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        ss = torch.empty(groups * c * 2, dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, lib.cstp_b16_bn_workspace_bytes(n, c, s, groups))
        with _span("bn_forward", (n, c, s, groups, res is not None, bool(relu), "bf16")):
            check(lib.cstp_b16_bn_forward_train(_stream(), x.data_ptr(), _ptr(res), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                                                ss.data_ptr(), n, c, s, groups, eps, momentum, 1 if relu else 0, ws.data_ptr(),
                                                ws.numel()), "cstp_b16_bn_forward_train")
        # ReLU without a residual: backward recomputes the mask from x with the affine table instead of re-reading y
        ctx.remask = bool(relu and res is None)
        ctx.save_for_backward(x, ss if ctx.remask else y, gamma, mean, invstd, ss)
        ctx.relu, ctx.groups, ctx.has_res = relu, groups, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y_or_ss, gamma, mean, invstd, ss = ctx.saved_tensors
        y = None if ctx.remask else y_or_ss
        dy = _req16(dy, "batch_norm grad_output")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        pg, pb = ctx.param_objs
        direct = ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _direct(pg) and _direct(pb)
        dgamma = pg.grad if direct else torch.empty_like(gamma)
        dbeta = pb.grad if direct else torch.empty_like(gamma)
        ws = _workspace(x.device, lib.cstp_b16_bn_workspace_bytes(n, c, s, ctx.groups))
        with _span("bn_backward", (n, c, s, ctx.groups, ctx.has_res, bool(ctx.relu), "bf16")):
            check(lib.cstp_b16_bn_backward(_stream(), x.data_ptr(), _ptr(y), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                           invstd.data_ptr(), ss.data_ptr(), dx.data_ptr(), _ptr(dres), dgamma.data_ptr(),
                                           dbeta.data_ptr(), n, c, s, ctx.groups, 1 if ctx.relu else 0, ws.data_ptr(), ws.numel(),
                                           1 if direct else 0), "cstp_b16_bn_backward")
        if direct:
            dgamma = dbeta = None
        if dres is not None and ctx.grad_join is not None:       # the residual tensor's other consumer adds its gradient to this
            dres = ctx.grad_join.contribute(lambda: dres, lambda buf: buf.add_(dres))
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


class _MaxPool3dB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        lib = _lib.load()
        x = _req16(x, "max_pool3d input")
        if x.dim() != 5:
            raise _lib.CstpError("max_pool3d expects [N, C, D, H, W], got %s" % (tuple(x.shape),))
        n, c, d, h, w = x.shape
        osz = tuple((sz + 2 * padding[i] - kernel[i]) // stride[i] + 1 for i, sz in enumerate((d, h, w)))
        y = torch.empty((n, c) + osz, dtype=torch.bfloat16, device=x.device)
        idx = torch.empty((n, c) + osz, dtype=torch.int32, device=x.device)
        check(lib.cstp_b16_maxpool3d_forward(_stream(), x.data_ptr(), y.data_ptr(), idx.data_ptr(), n * c, d, h, w,
                                             (ctypes.c_int32 * 3)(*kernel), (ctypes.c_int32 * 3)(*stride),
                                             (ctypes.c_int32 * 3)(*padding)), "cstp_b16_maxpool3d_forward")
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape), tuple(kernel), tuple(stride), tuple(padding))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        shape, kernel, stride, padding = ctx.geom
        dy = _req16(dy, "max_pool3d grad_output")
        dx = torch.empty(shape, dtype=torch.bfloat16, device=dy.device)
        n, c, d, h, w = shape
        check(lib.cstp_b16_maxpool3d_backward(_stream(), dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n * c, d, h, w,
                                              (ctypes.c_int32 * 3)(*kernel), (ctypes.c_int32 * 3)(*stride),
                                              (ctypes.c_int32 * 3)(*padding)), "cstp_b16_maxpool3d_backward")
        return dx, None, None, None


class _AvgPoolB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req16(x, "avgpool input")
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(lib.cstp_b16_avgpool_forward(_stream(), x.data_ptr(), y.data_ptr(), n * c, s), "cstp_b16_avgpool_forward")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _req(dy, "avgpool grad_output")
        dx = torch.empty(ctx.shape, dtype=torch.bfloat16, device=dy.device)
        n, c = ctx.shape[0], ctx.shape[1]
        check(lib.cstp_b16_avgpool_backward(_stream(), dy.data_ptr(), dx.data_ptr(), n * c, dx.numel() // (n * c)),
              "cstp_b16_avgpool_backward")
        return dx


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
class _ByolLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        x = _req(x, "byol x")
        y = _req(y, "byol y")
        b, f = x.shape
        loss = torch.empty(b, dtype=torch.float32, device=x.device)
        check(lib.cstp_byol_loss_forward(_stream(), x.data_ptr(), y.data_ptr(), loss.data_ptr(), b, f), "cstp_byol_loss_forward")
        ctx.save_for_backward(x, y)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        x, y = ctx.saved_tensors
        dloss = _req(dloss, "byol grad_output")
        dx = torch.empty_like(x)
        check(lib.cstp_byol_loss_backward(_stream(), x.data_ptr(), y.data_ptr(), dloss.data_ptr(), dx.data_ptr(), x.shape[0],
                                          x.shape[1]), "cstp_byol_loss_backward")
        return dx, None


def byol_regression_loss(x, y):
    """2 - 2*cos(x, y) per row; y is treated as a constant (the detached target projection)."""
    return _ByolLoss.apply(x, y.detach())


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        x = _req(x, "normalize input")
        rows, f = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.cstp_l2_normalize_forward(_stream(), x.data_ptr(), y.data_ptr(), norm.data_ptr(), rows, f, eps),
              "cstp_l2_normalize_forward")
        ctx.save_for_backward(y, norm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        y, norm = ctx.saved_tensors
        dy = _req(dy, "normalize grad_output")
        dx = torch.empty_like(y)
        check(lib.cstp_l2_normalize_backward(_stream(), y.data_ptr(), norm.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                             y.shape[0], y.shape[1], ctx.eps), "cstp_l2_normalize_backward")
        return dx, None


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1) for 2-D x (r21d_byol.py:396)."""
    if x.dim() != 2:
        raise _lib.CstpError("l2_normalize expects [rows, features], got %s" % (tuple(x.shape),))
    return _L2Normalize.apply(x, float(eps))


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lib = _lib.load()
        logits = _req(logits, "cross_entropy logits")
        if labels.dtype != torch.int64 or not labels.is_cuda:
            raise _lib.CstpError("cross_entropy labels must be int64 on the HIP device")
        labels = labels.contiguous()
        b, k = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        check(lib.cstp_cross_entropy_forward(_stream(), logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), b, k),
              "cstp_cross_entropy_forward")
        ctx.save_for_backward(logits, labels)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, labels = ctx.saved_tensors
        dloss = _req(dloss.reshape(1), "cross_entropy grad_output")
        dl = torch.empty_like(logits)
        check(lib.cstp_cross_entropy_backward(_stream(), logits.data_ptr(), labels.data_ptr(), dloss.data_ptr(), dl.data_ptr(),
                                              logits.shape[0], logits.shape[1]), "cstp_cross_entropy_backward")
        return dl, None


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() (mean reduction)."""
    return _CrossEntropy.apply(logits, labels)


class _NTXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reps, temperature):
        lib = _lib.load()
        reps = _req(reps, "ntxent representations")
        two_n, f = reps.shape
        nbytes = lib.cstp_ntxent_workspace_bytes(two_n, f)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=reps.device)  # private: lives until backward
        loss = torch.empty(1, dtype=torch.float32, device=reps.device)
        check(lib.cstp_ntxent_forward(_stream(), reps.data_ptr(), loss.data_ptr(), two_n, f, temperature, ws.data_ptr(),
                                      ws.numel()), "cstp_ntxent_forward")
        ctx.save_for_backward(reps, ws)
        ctx.temperature = temperature
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        reps, ws = ctx.saved_tensors
        dloss = _req(dloss.reshape(1), "ntxent grad_output")
        dreps = torch.empty_like(reps)
        check(lib.cstp_ntxent_backward(_stream(), reps.data_ptr(), dloss.data_ptr(), dreps.data_ptr(), reps.shape[0],
                                       reps.shape[1], ctx.temperature, ws.data_ptr(), ws.numel()), "cstp_ntxent_backward")
        return dreps, None


def ntxent(reps, temperature):
    """NT-Xent over reps = cat(zjs, zis) [2N, F] (loss/NTXent.py:46-62)."""
    return _NTXent.apply(reps, float(temperature))


# ----------------------------------------------------------------------------------------------
# flat-arena utilities (no autograd)
# ----------------------------------------------------------------------------------------------
def ema_update_(target: torch.Tensor, online: torch.Tensor, m: float) -> None:
    lib = _lib.load()
    assert target.is_cuda and target.is_contiguous() and online.is_contiguous() and target.numel() == online.numel()
    check(lib.cstp_ema_update(_stream(), target.data_ptr(), online.data_ptr(), target.numel(), float(m)), "cstp_ema_update")


def grad_sumsq(g: torch.Tensor, out: torch.Tensor) -> None:
    lib = _lib.load()
    ws = _workspace(g.device, 8192)
    check(lib.cstp_sumsq(_stream(), g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), ws.numel()), "cstp_sumsq")


def clip_coef(sumsq: torch.Tensor, max_norm: float, coef: torch.Tensor, norm_out: Optional[torch.Tensor]) -> None:
    lib = _lib.load()
    check(lib.cstp_clip_coef(_stream(), sumsq.data_ptr(), float(max_norm), coef.data_ptr(), _ptr(norm_out)), "cstp_clip_coef")


def sgd_step_(p, g, buf, lr: torch.Tensor, momentum: float, weight_decay: float, coef: Optional[torch.Tensor],
              first_step: bool, write_back_grad: bool = True) -> None:
    lib = _lib.load()
    check(lib.cstp_sgd_step(_stream(), p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), lr.data_ptr(), float(momentum),
                            float(weight_decay), _ptr(coef), 1 if first_step else 0, 1 if write_back_grad else 0),
          "cstp_sgd_step")


def adam_step_(p, g, exp_avg, exp_avg_sq, lr: torch.Tensor, beta1: float, beta2: float, eps: float, weight_decay: float,
               decoupled: bool, step: int) -> None:
    lib = _lib.load()
    check(lib.cstp_adam_step(_stream(), p.data_ptr(), g.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), p.numel(),
                             lr.data_ptr(), float(beta1), float(beta2), float(eps), float(weight_decay),
                             1 if decoupled else 0, int(step)), "cstp_adam_step")
