"""N > 1 path with the HIP kernels as the per-shard compute: two ranks share
GPU 0 of the box (one process each, ``gloo`` for the exchange because RCCL refuses two ranks on one
device), every rank owns a contiguous block of Gauss points with its state resident in its own
``HIPMaterial`` handle, integrates the whole load/unload history and all-gathers stress, tangent and
internal state variables; rank 0's reassembled arrays must equal the single-process oracle on the
full batch.  Also: ``python bench.py --gpus 2`` starts its own ranks (no external launcher)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history, to_device

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _worker(rank, world, port, n, law, q):
    import torch
    import torch.distributed as dist

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from dolfinx_materials_amd.sharding import ShardPlan, allgather_rows, allgather_rows_p2p, allgather_tangent

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = ShardPlan(n, world)
        lo, hi = plan.range(rank)
        sig0 = SIG0_LIN if law == "linear" else SIG0_V
        hard = jm.LinearHardening(SIG0_LIN, H_LIN) if law == "linear" else jm.VoceHardening(SIG0_V, SIGU_V, B_V)
        beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), hard)
        mat, cmat, pmat = JAXMaterial(beh, device=0), JAXMaterial(beh, device=0, tangent_layout="coef"), JAXMaterial(beh, device=0, tangent_layout="pack4")
        for m_ in (mat, cmat, pmat):
            m_.set_data_manager(hi - lo)
        out = []
        for eps in j2_history(n, seed=77, sig0=sig0):  # same global batch on every rank; each takes its block
            sig, isv, ct = mat.integrate(eps[lo:hi])
            assert mat.last_stats["n_nan"] == 0 and mat.last_stats["n_not_converged"] == 0
            g_sig = allgather_rows(torch.from_numpy(np.array(sig)), plan)
            g_ct = allgather_rows(torch.from_numpy(np.array(ct).reshape(-1, 36)), plan)
            g_isv = allgather_rows_p2p(torch.from_numpy(np.array(isv)), plan)
            assert torch.equal(g_ct, allgather_rows_p2p(torch.from_numpy(np.array(ct).reshape(-1, 36)), plan))
            # the third schedule: the nine coefficients per point on the wire, blocks rebuilt on every rank ...
            _, _, coef = cmat.integrate(eps[lo:hi])
            for p2p in (False, True):
                g_ct9 = allgather_tangent(torch.from_numpy(np.array(coef)), plan, p2p=p2p)
                assert (g_ct9 - g_ct).abs().max() <= 1e-15 * g_ct.abs().max()
            # ... and the 32 B/point form: (c1, c2, c3, w), the flow direction rebuilt from the gathered stress
            _, _, pack = pmat.integrate(eps[lo:hi])
            assert np.array_equal(np.array(pack)[:, :3], np.array(coef)[:, :3])
            g_ct4 = allgather_tangent(torch.from_numpy(np.array(pack)), plan, p2p=True, flux_all=g_sig)
            assert (g_ct4 - g_ct).abs().max() <= 1e-15 * g_ct.abs().max()
            out.append((g_sig.numpy(), g_isv.numpy(), g_ct.numpy()))
            for m_ in (mat, cmat, pmat):
                m_.data_manager.update()
        if rank == 0:
            q.put(out)
        for m_ in (mat, cmat, pmat):
            m_.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,law", [(2, 4096, "linear"), (2, 10_001, "voce"), (8, 10_003, "voce"), (8, 4096, "linear")])
def test_ranks_with_hipmaterial_per_shard_match_single_process_oracle(gpu_available, world, n, law):
    """Two ranks, and the eight of one node with a ragged 8-way plan (10 003 points: blocks of 1251 and 1250), all on
    GPU 0 of the box: collective, point-to-point schedule with 7 peers and the coefficient gather."""
    if not gpu_available:
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    from oracle import constitutive_np as onp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, law, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = _collect(q, procs)
    sig0 = SIG0_LIN if law == "linear" else SIG0_V
    hard = onp.LinearHardening(SIG0_LIN, H_LIN) if law == "linear" else onp.VoceHardening(SIG0_V, SIGU_V, B_V)
    epsp, p_ = np.zeros((n, 6)), np.zeros(n)
    for eps, (sig, isv, ct) in zip(j2_history(n, seed=77, sig0=sig0), got):
        ref = onp.j2_update(eps, epsp, p_, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * sig0
        for a, b in ((sig, ref["sig"]), (ct, ref["Ct"].reshape(n, 36)), (isv[:, 0], ref["p"]), (isv[:, 1:], ref["epsp"])):
            err = np.abs(a[safe] - b[safe]).max() / max(np.abs(b).max(), 1e-300)
            assert err < 1e-12, err  # contract: rtol 1e-6 plastic (BASELINE.json north_star)
        epsp, p_ = ref["epsp"], ref["p"]


def test_bench_starts_its_own_ranks(gpu_available):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py launches the two
    ranks itself before any GPU call (debug share mode: both on GPU 0, gloo) and rank 0 prints ONE line."""
    if not gpu_available:
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--points", "300000",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-other-laws", "--gather-steps", "1", "--cfg3-points", "20000"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"]["frac"] > 0 and out["roofline"]["kernel_ms"] > 0
    pg = out["process_group"]
    assert pg["ranks_in_group"] == 2 and pg["ranks_counted_by_all_reduce"] == 2 and pg["launcher"] == "self"
    assert out["gather_inclusive"]["value"] > 0 and out["gather_inclusive"]["p2p_schedule"]["value"] > 0
    assert out["value"] > out["gather_inclusive"]["value"]


def test_bench_under_the_drivers_launcher(gpu_available):
    """The driver's N > 1 command verbatim -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks sharing GPU 0 (DXM_BENCH_SHARE_GPU=1: gloo): RANK /
    LOCAL_RANK / WORLD_SIZE come from the launcher, rank 0 prints the one line."""
    if not gpu_available:
        pytest.skip("no GPU")
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    # (HSA_ENABLE_IPC_MODE_LEGACY is taken OUT of the environment: the driver's launcher will not set it; bench.py::main does, for every rank)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(DXM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--points", "200000", "--cfg3-points", "20000", "--cpu-sample", "100000", "--gather-steps", "1", "--settle-seconds", "0.1"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["value"] > 0
    pg = out["process_group"]
    assert pg["ranks_counted_by_all_reduce"] == 2 and pg["launcher"] == "torch.distributed.run" and pg["share_gpu_debug_mode"]
    assert pg["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one HBM budget line per rank on stderr, before anything is allocated (headline + cfg3 block)
    budget = [json.loads(l.split("] ", 1)[1]) for l in r.stderr.splitlines() if l.startswith("[bench.py hbm budget] ")]
    assert sorted(b["rank"] for b in budget) == [0, 1] and all(set(b["blocks_GB"]) == {"headline", "cfg3"} for b in budget)
    assert all(0 < b["largest_block_GB"] < 0.9 * b["hbm_free_GB"] for b in budget)
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0 and "error" not in out["cfg3"]


@pytest.mark.parametrize("world", [4, 8])
def test_bench_line_of_the_drivers_scaling_sweep_on_one_gpu(gpu_available, world):
    """What the driver's 1 -> 8 sweep runs is `python bench.py --gpus N`: here the same command for N = 4 and 8 in the debug
    share mode (all ranks on GPU 0, gloo; N = 2 above), small batches, so that no N of the sweep is an unexercised path: N ranks
    counted by the all-reduce, the cfg 2 weak-scaling headline with `roofline`, `cpu_baseline` and its gather legs, the `box`
    block of rank 0, and the `cfg3` block (Voce, its own points per rank) with the three reassembly schedules at >= 10 steps."""
    if not gpu_available:
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--points", "200000",
                        "--cfg3-points", "25000", "--steps", "6", "--warmup", "2", "--cpu-sample", "100000", "--gather-steps", "1", "--settle-seconds", "0.1"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["law"] == "j2_linear" and out["value"] > 0 and out["scaling"] == "weak"
    assert out["steps"] == 6 and out["warmup"] == 2 and out["dtype"] == "f64" and out["vs_baseline"] is None
    pg = out["process_group"]
    assert pg["ranks_in_group"] == world and pg["ranks_counted_by_all_reduce"] == world and pg["launcher"] == "self"
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["kernel"].startswith("small_strain_kernel<1")
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    assert abs(rf["achieved"] - 496 * 200000 / (rf["kernel_ms"] * 1e-3) / 1e9) <= 0.01 * rf["achieved"]
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] >= 1
    assert "before" in out["box"] and out["box"]["during_timed_steps"] is not None
    g2 = out["gather_inclusive"]
    assert g2["value"] > 0 and g2["p2p_schedule"]["value"] > 0 and g2["coefficient_gather"]["value"] > 0
    c3 = out["cfg3"]
    assert "error" not in c3 and c3["points_per_gpu"] == 25000 and c3["points_total"] == 25000 * world and c3["value"] > 0
    assert "Voce" in c3["workload"] and c3["kernel"].startswith("small_strain_kernel<2")
    g = c3["gather_inclusive"]
    assert g["steps"] >= 10 and g["value"] > 0 and g["p2p_schedule"]["value"] > 0 and g["coefficient_gather"]["value"] > 0


@pytest.mark.parametrize("n", [64, 1000, 250_007])
def test_coefficient_gather_payload_is_rebuilt_bit_for_bit_on_the_device(gpu_available, n):
    """`sharding.allgather_tangent`: ranks exchange the 72 B/point coefficient form and rebuild the (N,36) blocks with
    `dxm_expand_tangent_device`; here the rebuild alone (single process), against the full-layout kernel."""
    if not gpu_available:
        pytest.skip("no GPU")
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd import _lib
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    dev = torch.device("cuda:0")
    beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_V, SIGU_V, B_V))
    full, coef, pack = JAXMaterial(beh), JAXMaterial(beh, tangent_layout="coef"), JAXMaterial(beh, tangent_layout="pack4")
    for m_ in (full, coef, pack):
        m_.set_data_manager(n)
    st = torch.cuda.current_stream().cuda_stream
    f = torch.empty((n, 6), dtype=torch.float64, device=dev)
    c36, c9, out = (torch.empty((n, k), dtype=torch.float64, device=dev) for k in (36, 9, 36))
    c4 = torch.empty((n, 4), dtype=torch.float64, device=dev)
    lib = _lib.load()
    for eps in j2_history(n, seed=8, sig0=SIG0_V):
        g = to_device(eps, dev)   # through a page-locked staging tensor
        full.integrate_device(g.data_ptr(), f.data_ptr(), c36.data_ptr(), st)
        coef.integrate_device(g.data_ptr(), f.data_ptr(), c9.data_ptr(), st)
        out.fill_(float("nan"))
        _lib.check(lib.dxm_expand_tangent_device(c9.data_ptr(), n, out.data_ptr(), 0, st or None), lib)
        torch.cuda.synchronize()
        assert torch.equal(out, c36)
        # the 32 B/point form + the stress of the same update
        pack.integrate_device(g.data_ptr(), f.data_ptr(), c4.data_ptr(), st)
        out.fill_(float("nan"))
        _lib.check(lib.dxm_expand_tangent_pack4_device(f.data_ptr(), c4.data_ptr(), n, out.data_ptr(), 0, st or None), lib)
        torch.cuda.synchronize()
        assert torch.equal(out, c36) and torch.equal(c4[:, :3], c9[:, :3])
        for m_ in (full, coef, pack):
            m_.data_manager.update()
    assert full.stats()[1]["n_plastic"] == 0 and float(c9[:, 2].abs().max()) == 0.0   # last increment unloads


def test_gather_legs_run_through_rccl_itself_in_a_group_of_one(gpu_available):
    """RCCL refuses two ranks on one device, so the 2-rank tests above use gloo; here the SAME calls (all_gather_into_tensor
    on device tensors, the p2p schedule, the coefficient gather + dxm_expand_tangent_device) go through the `nccl` backend
    in a process group of one rank -- what a 1-GPU box can verify of the N > 1 data path before an 8-GPU node runs it."""
    if not gpu_available:
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DXM_BENCH_SHARE_GPU",
                                                            "HSA_ENABLE_IPC_MODE_LEGACY")}   # (bench.py::main sets it itself)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-rank-group", "--points", "300000",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-other-laws", "--no-host-path", "--no-live-traffic",
                        "--gather-steps", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    pg, g = out["process_group"], out["gather_inclusive"]
    assert pg["backend"] == "nccl" and pg["ranks_in_group"] == 1 and pg["ranks_counted_by_all_reduce"] == 1
    assert pg["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "error" not in g and g["value"] > 0
    assert "error" not in g["p2p_schedule"] and g["p2p_schedule"]["value"] > 0
    assert "error" not in g["coefficient_gather"] and g["coefficient_gather"]["value"] > 0
    assert out["n_gpus"] == 1 and out["value"] > 0   # (in place and with one rank the gather legs cost next to nothing)
