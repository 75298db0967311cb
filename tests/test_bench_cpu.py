"""CPU: the parts of bench.py that do not need a GPU (synthetic history, CPU-baseline leg)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import constitutive_np as onp  # noqa: E402


def test_history_is_seeded_and_has_the_documented_plastic_fractions():
    """SURVEY.md 8(d) cfg 2: proportional loading eps_k = (k/3) eps_hat, then unloading to 0.5 eps_hat;
    a majority of the points are plastic at the peak."""
    h1, h2 = bench.history(20000, 1234), bench.history(20000, 1234)
    assert all(np.array_equal(a, b) for a, b in zip(h1, h2))
    assert not np.array_equal(h1[2], bench.history(20000, 1235)[2])
    assert np.allclose(h1[0] * 3, h1[2]) and np.allclose(h1[1] * 1.5, h1[2]) and np.allclose(h1[3] * 2, h1[2])
    hard = onp.LinearHardening(bench.SIG0, bench.H)
    epsp, p, fr = np.zeros((20000, 6)), np.zeros(20000), []
    for eps in h1:
        r = onp.j2_update(eps, epsp, p, bench.E, bench.NU, hard)
        fr.append(r["plastic"].mean())
        epsp, p = r["epsp"], r["p"]
    assert 0.5 < fr[1] < 0.65 and 0.65 < fr[2] < 0.8 and fr[3] == 0.0


def test_cpu_baseline_leg_runs_and_reports_threads():
    out = bench.cpu_baseline(20000, 1234, budget_s=1.0)
    assert out["kind"] == "port" and out["unit"] == "Mpoints/s" and out["value"] > 0
    assert 1 <= out["cores"] <= (os.cpu_count() or 1) and "20000 points" in out["sample"]


def test_self_launcher_propagates_a_failing_rank_without_hanging():
    """`python bench.py --gpus 2` starts its own ranks; here (no GPU) every rank fails its GPU assertion and the
    launcher must come back with a non-zero code instead of leaving ranks waiting in the rendezvous."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"),
                        "--gpus", "2", "--points", "1000", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch

    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_source_hash_ignores_comments_and_matches_the_stamped_traffic_file():
    """`bench.source_hash`: the identity of the code a measurement belongs to -- comments and white space do not change it, code does --
    and `profiles/pmc_traffic.json` carries the hash of the shipped sources."""
    import json
    import os

    import bench

    code = 'int a = 1; // one\n/* block\n comment */ const char* s = "// kept /* kept */";\n\n   double  b = a /2.0 ;'
    same = 'int a = 1;\nconst char* s = "// kept /* kept */"; double b = a /2.0 ; // trailing'
    assert bench._code_only(code) == bench._code_only(same)
    assert bench._code_only(code) != bench._code_only(code.replace("2.0", "3.0"))
    stamped = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")))
    assert stamped["source_hash"] == bench.source_hash()
