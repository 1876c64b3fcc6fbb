"""NT-Xent contrastive head (mirror of /root/reference/loss/NTXent.py:5-62) with the
data-parallel extension north_star asks for: negatives all-gathered across ranks on RCCL.

The reference builds ``NTXentLoss(batch_size=opts.batch_size)`` with the GLOBAL batch
(main_byol.py:191-197), so the only shape-consistent input under DDP is the all-gathered
embedding matrix.  ``forward(zis, zjs)`` keeps the reference signature; with an initialised
process group and ``gather=True`` the local [B_local, F] embeddings are gathered to
[B_global, F] first.  The gather carries gradient only for this rank's slice (every rank
evaluates the identical global loss), so the caller multiplies the loss by world_size to
undo DDP's gradient averaging -- ``ddp_scale`` holds that factor.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


def all_gather_with_grad(z: torch.Tensor) -> torch.Tensor:
    """cat over ranks of z (rank-major); gradient flows into the local slice only."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return z
    world, rank, b = dist.get_world_size(), dist.get_rank(), z.shape[0]
    # ONE collective straight into the [world * B, F] result (no per-rank allocations, no cat); the local slice is then put
    # back as the live tensor so that the gradient reaches this rank's embeddings
    out = torch.empty((world * b,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
    zc = z.detach().contiguous()
    # The form of the collective is chosen ONCE FROM THE BACKEND -- identically on every rank -- and real errors propagate.
    # (Round-3 ADVICE: a try / except around the tensor form let ONE rank fall back to the list form after, say, an
    # out-of-memory error while its peers had completed the tensor form: mismatched collectives hang the job.)
    if _tensor_form(z):
        dist.all_gather_into_tensor(out, zc)
    else:
        dist.all_gather(list(out.split(b, dim=0)), zc)
    return _PutLocal.apply(out, z, rank * b)


def _tensor_form(z: torch.Tensor) -> bool:
    """all_gather_into_tensor where the backend that serves ``z``'s device has it (RCCL / NCCL: yes; gloo: the list form)."""
    backend = dist.get_backend()
    if ":" in backend:              # "cpu:gloo,cuda:nccl": the entry of the tensor's device type
        kinds = dict(part.split(":") for part in backend.split(","))
        backend = kinds.get("cuda" if z.is_cuda else "cpu", "gloo")
    return backend.lower() in ("nccl", "rccl")


class _PutLocal(torch.autograd.Function):
    """gathered[off : off + B] <- z, with d(out)/dz = that slice of the incoming gradient (the other rows are constants)."""

    @staticmethod
    def forward(ctx, gathered, z, off):
        ctx.off, ctx.b = off, z.shape[0]
        return gathered.view_as(gathered)      # (already holds z's values in its slice: the all-gather wrote them)

    @staticmethod
    def backward(ctx, g):
        return None, g[ctx.off:ctx.off + ctx.b], None


class NTXentLoss(torch.nn.Module):
    def __init__(self, device, batch_size, temperature, use_cosine_similarity=True, gather=True, kernel=None):
        """``kernel(reps [2N, F], temperature) -> scalar``: the HIP NT-Xent kernels by default (ops.ntxent); injectable so
        that the gather / scale logic can be driven on CPU (tests/test_dist_gloo.py)."""
        super().__init__()
        self._kernel = kernel if kernel is not None else ops.ntxent
        if not use_cosine_similarity:
            raise NotImplementedError("the CSTP driver only builds the cosine-similarity variant (main_byol.py:195)")
        self.batch_size, self.temperature, self.device, self.gather = batch_size, temperature, device, gather

    @property
    def ddp_scale(self) -> float:
        if self.gather and dist.is_available() and dist.is_initialized():
            return float(dist.get_world_size())
        return 1.0

    def forward(self, zis, zjs):
        if self.gather:
            zis, zjs = all_gather_with_grad(zis), all_gather_with_grad(zjs)
        if zis.shape[0] != self.batch_size or zjs.shape[0] != self.batch_size:
            # the reference fails here too: its mask is built for 2*batch_size rows (NTXent.py:23-29,57)
            raise RuntimeError("NTXentLoss was built for batch_size=%d but got %d embeddings"
                               % (self.batch_size, zis.shape[0]))
        return self._kernel(torch.cat([zjs, zis], dim=0), self.temperature)
