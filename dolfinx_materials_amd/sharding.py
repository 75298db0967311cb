"""Sharding of a Gauss-point batch over the GPUs of one node (one process per GPU).

The constitutive update has no cross-point dependency (reference ``jaxmat.py:147-151`` is a
``vmap``; ``generic.py:77-79`` a loop), so rank ``r`` owns the contiguous block
``[bounds[r], bounds[r+1])`` of the QuadratureMap ordering (cell-major, ``quadrature_map.py:255-260``)
and keeps that block's state resident on its GPU.  No collective is needed for the update itself;
``allgather_rows`` reassembles per-rank ``(n_r, dim)`` outputs (stress, tangent) into the full
``(N, dim)`` array on every rank -- RCCL over xGMI when the process group backend is ``nccl``,
``gloo`` in the CPU tests.

In a multi-rank dolfinx run each MPI rank only consumes its own points
(``quadrature_map.py:66-70``), so the gather is only for the single-consumer scenario.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardPlan:
    n_total: int
    world_size: int

    @property
    def bounds(self):
        """Balanced contiguous partition: the first ``n_total % world`` ranks get one more point."""
        q, r = divmod(self.n_total, self.world_size)
        b = [0]
        for k in range(self.world_size):
            b.append(b[-1] + q + (1 if k < r else 0))
        return b

    def range(self, rank):
        b = self.bounds
        return b[rank], b[rank + 1]

    def count(self, rank):
        lo, hi = self.range(rank)
        return hi - lo

    @property
    def max_count(self):
        return max(self.count(r) for r in range(self.world_size))

    def local_view(self, full: torch.Tensor, rank: int) -> torch.Tensor:
        """Rank ``rank``'s rows of a gathered ``(n_total, dim)`` array.  A kernel that writes its output THERE
        (``integrate_device(..., flux_ptr=view.data_ptr())``) makes the gathers below in-place: no local copy."""
        lo, hi = self.range(rank)
        return full[lo:hi]


def _is_own_block(local: torch.Tensor, out: torch.Tensor | None, lo: int, hi: int) -> bool:
    return out is not None and hi > lo and local.is_contiguous() and local.data_ptr() == out[lo:hi].data_ptr()


def allgather_rows(local: torch.Tensor, plan: ShardPlan, out: torch.Tensor | None = None,
                   scratch: torch.Tensor | None = None, group=None) -> torch.Tensor:
    """All-gather row blocks of unequal length into ``out`` of shape ``(n_total, dim)``.

    Equal shards use one ``all_gather_into_tensor`` straight into ``out`` (a single large
    collective: xGMI is point-to-point, so few big messages beat many small ones); when ``local`` already IS this
    rank's rows of ``out`` (:meth:`ShardPlan.local_view`) that is the in-place form of the collective and nothing is
    copied locally.  Ragged shards are padded to ``max_count`` rows in ``scratch`` and compacted afterwards.
    """
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert world == plan.world_size
    dim = local.shape[1]
    assert local.shape[0] == plan.count(rank)
    if out is None:
        out = torch.empty((plan.n_total, dim), dtype=local.dtype, device=local.device)
    if plan.n_total % world == 0:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    m = plan.max_count
    if scratch is None:
        scratch = torch.empty((world * m, dim), dtype=local.dtype, device=local.device)
    padded = torch.zeros((m, dim), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    dist.all_gather_into_tensor(scratch, padded, group=group)
    b = plan.bounds
    for r in range(world):
        out[b[r] : b[r + 1]] = scratch[r * m : r * m + (b[r + 1] - b[r])]
    return out


def allgather_rows_p2p(local: torch.Tensor, plan: ShardPlan, out: torch.Tensor | None = None, group=None) -> torch.Tensor:
    """Same result as :func:`allgather_rows`, scheduled as one batch of point-to-point transfers:
    every rank sends its block to each of the other ranks and receives theirs straight into the
    destination rows (``batch_isend_irecv``).  xGMI is a full mesh of point-to-point links
    (7 x ~153 GB/s per GPU): with all ``world - 1`` transfers of a rank in flight at once every
    link carries exactly one block, whereas a ring all-gather is bound by one link for
    ``world - 1`` sequential steps.  Ragged blocks need no padding here."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert world == plan.world_size
    dim = local.shape[1]
    lo, hi = plan.range(rank)
    assert local.shape[0] == hi - lo
    if out is None:
        out = torch.empty((plan.n_total, dim), dtype=local.dtype, device=local.device)
    if not _is_own_block(local, out, lo, hi):
        out[lo:hi] = local
    src = local.contiguous()
    ops = []
    b = plan.bounds
    for shift in range(1, world):
        dst = (rank + shift) % world          # staggered so that no two ranks target the same peer first
        frm = (rank - shift) % world
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, src, dst, group=group))
        if b[frm + 1] > b[frm]:
            ops.append(dist.P2POp(dist.irecv, out[b[frm] : b[frm + 1]], frm, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def allgather_tangent(coef_local: torch.Tensor, plan: ShardPlan, out: torch.Tensor | None = None,
                      coef_all: torch.Tensor | None = None, group=None, p2p: bool = False,
                      flux_all: torch.Tensor | None = None) -> torch.Tensor:
    """Reassemble the full ``(N, 36)`` J2 tangent on every rank from per-rank COEFFICIENT blocks: ``(n_r, 9)`` (materials
    created with ``tangent_layout="coef"``: 72 instead of 288 B/point cross the links) or ``(n_r, 4)`` (``"pack4"``: 32
    B/point; needs ``flux_all``, the already gathered ``(N, 6)`` stress of the same update, from which the flow direction is
    rebuilt), then every rank rebuilds the blocks locally with ``dxm_expand_tangent[_pack4]_device`` -- the update kernel's own
    expression, bit-identical to gathering full blocks.  xGMI is the bound of the gather-inclusive figure (SURVEY.md section
    8(e): 27 ms of link time against 2 ms of compute per shard at cfg 3), HBM is not: the rebuild costs 360-410 B/point of local
    traffic (~6 ms for 1e8 points) and saves 216 / 256 B/point on the links.  CPU tensors (the gloo tests) are rebuilt with numpy."""
    width = coef_local.shape[1]
    if width not in (4, 9):
        raise ValueError("coefficient blocks are (n, 9) ('coef') or (n, 4) ('pack4')")
    if width == 4 and flux_all is None:
        raise ValueError("the 'pack4' form needs flux_all, the gathered stress of the same update")
    gather = allgather_rows_p2p if p2p else allgather_rows
    kw = {} if p2p else {"scratch": None}
    coef_all = gather(coef_local, plan, out=coef_all, group=group, **kw)
    if out is None:
        out = torch.empty((plan.n_total, 36), dtype=coef_local.dtype, device=coef_local.device)
    if coef_all.is_cuda:
        from . import _lib

        lib = _lib.load()
        st = torch.cuda.current_stream(coef_all.device).cuda_stream or None
        dev = coef_all.device.index or 0
        if width == 9:
            _lib.check(lib.dxm_expand_tangent_device(coef_all.data_ptr(), plan.n_total, out.data_ptr(), dev, st), lib)
        else:
            _lib.check(lib.dxm_expand_tangent_pack4_device(flux_all.data_ptr(), coef_all.data_ptr(), plan.n_total, out.data_ptr(), dev, st), lib)
    else:
        from .conventions import tangent_from_coefficients, tangent_from_pack4

        full = tangent_from_coefficients(coef_all.numpy()) if width == 9 else tangent_from_pack4(flux_all.numpy(), coef_all.numpy())
        out.copy_(torch.from_numpy(full.reshape(-1, 36)))
    return out
