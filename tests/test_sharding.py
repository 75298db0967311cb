"""N > 1 path on CPU: two gloo ranks own contiguous point blocks, update them independently
(here with the oracle as the per-shard compute -- test infrastructure) and all-gather stress and
tangent; the result must equal the single-process full-batch update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dolfinx_materials_amd.sharding import ShardPlan, allgather_rows, allgather_rows_p2p, allgather_tangent
from oracle import constitutive_np as onp

from helpers import E, NU, SIG0_LIN, H_LIN, j2_history


def test_shard_plan_partitions():
    for n, w in [(10, 2), (11, 2), (7, 8), (0, 4), (100_000_001, 8)]:
        p = ShardPlan(n, w)
        b = p.bounds
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        counts = [p.count(r) for r in range(w)]
        assert sum(counts) == n and max(counts) - min(counts) <= 1 and p.max_count == max(counts)


def _collect(q, procs, timeout=600):
    """Rank 0's result; fails as soon as any rank has died instead of waiting for the queue's timeout."""
    import queue as _queue
    import time as _time

    t0, got = _time.time(), None
    while got is None:
        try:
            got = q.get(timeout=1.0)
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or _time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"ranks exited with {dead}" if dead else "timed out waiting for rank 0")
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return got


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = ShardPlan(n, world)
        lo, hi = plan.range(rank)
        eps = j2_history(n, seed=77)[2]  # same global batch on every rank; each takes its block
        hard = onp.LinearHardening(SIG0_LIN, H_LIN)
        r = onp.j2_update(eps[lo:hi], np.zeros((hi - lo, 6)), np.zeros(hi - lo), E, NU, hard)
        if hi == lo:   # a rank without points (fewer points than ranks): empty blocks of the right widths
            r = dict(sig=np.zeros((0, 6)), Ct=np.zeros((0, 6, 6)), plastic=np.zeros(0, dtype=bool), p=np.zeros(0))
        sig = allgather_rows(torch.from_numpy(r["sig"]), plan)
        ct = allgather_rows(torch.from_numpy(r["Ct"].reshape(-1, 36)), plan)
        # the point-to-point schedule must reassemble the very same arrays
        sig2 = allgather_rows_p2p(torch.from_numpy(r["sig"]), plan)
        ct2 = allgather_rows_p2p(torch.from_numpy(r["Ct"].reshape(-1, 36)), plan)
        assert torch.equal(sig, sig2) and torch.equal(ct, ct2)
        # in-place form: the producer writes into its own rows of the gathered array (ShardPlan.local_view)
        for gather in (allgather_rows, allgather_rows_p2p):
            full = torch.full((n, 36), float("nan"), dtype=torch.float64)
            mine = plan.local_view(full, rank)
            mine.copy_(torch.from_numpy(r["Ct"].reshape(-1, 36)))
            got = gather(mine, plan, out=full)
            assert got.data_ptr() == full.data_ptr() and torch.equal(full, ct)
        # coefficient form: 9 instead of 36 doubles per point on the wire, rebuilt locally
        one = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
        nrm = np.zeros((hi - lo, 6))                            # (c1, c2, c3, n) from the oracle's pieces
        from oracle.constitutive_np import lame
        lmbda, mu = lame(E, NU)
        eel = eps[lo:hi]
        se = 2 * mu * (eel - eel[:, :3].sum(1)[:, None] / 3 * one)
        seq = np.sqrt(1.5 * (se * se).sum(1))
        pl = r["plastic"]
        nrm[pl] = 1.5 * se[pl] / seq[pl, None]
        dp = r["p"]
        beta = np.where(pl, dp / np.where(pl, seq, 1.0), 0.0)
        gamma = np.where(pl, 1.0 / (H_LIN + 3 * mu), 0.0)
        coef = np.column_stack([lmbda + 2 * mu * mu * beta, 2 * mu - 6 * mu * mu * beta, 4 * mu * mu * (beta - gamma), nrm])
        for p2p in (False, True):
            full = allgather_tangent(torch.from_numpy(coef), plan, p2p=p2p)
            assert (full - ct).abs().max() <= 1e-12 * ct.abs().max()
        # 32 B/point form: (c1, c2, c3, w) with n = dev(stress) w rebuilt from the gathered stress
        sdev = r["sig"] - r["sig"][:, :3].sum(1)[:, None] / 3 * one
        seq_f = np.sqrt(1.5 * (sdev * sdev).sum(1))
        w = np.where(pl, 1.5 / np.where(pl, seq_f, 1.0), 0.0)
        pack = np.column_stack([coef[:, :3], w])
        for p2p in (False, True):
            full = allgather_tangent(torch.from_numpy(pack), plan, p2p=p2p, flux_all=sig)
            assert (full - ct).abs().max() <= 1e-12 * ct.abs().max()
        if rank == 0:
            q.put((sig.numpy(), ct.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 101), (8, 64), (8, 10_003), (8, 5)])
def test_sharded_update_and_allgather_matches_single_process(world, n):
    """Two ranks and the eight of one node (even, ragged, and fewer points than ranks: some ranks own nothing): the
    collective, the point-to-point schedule with 7 peers, the in-place forms and the coefficient gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    sig, ct = _collect(q, procs)
    eps = j2_history(n, seed=77)[2]
    ref = onp.j2_update(eps, np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))
    assert np.array_equal(sig, ref["sig"]) and np.array_equal(ct, ref["Ct"].reshape(n, 36))
