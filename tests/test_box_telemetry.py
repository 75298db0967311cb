"""tools/box_telemetry.py on a fabricated sysfs tree (no GPU here): the card of THIS process is the one whose render node opens, not
card0 of the node (the mistake of the first survey call); DPM files, hwmon, KFD topology and the firmware's gpu_metrics (through a
fake rocm-smi) are parsed; the sampler thread and the accumulator deltas behave; every missing file is a null, never an exception.
The module sits on bench.py's critical path: nothing it does may raise."""
import json
import os
import stat
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import box_telemetry as bt  # noqa: E402


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def _card(root, name, pci, minor, uid, busy, sclk_line):
    pcidir = os.path.join(root, "devices", pci)
    dev = os.path.join(root, "drm", name, "device")
    os.makedirs(os.path.join(root, "drm", name), exist_ok=True)
    os.makedirs(pcidir, exist_ok=True)
    os.symlink(pcidir, dev)
    os.makedirs(os.path.join(pcidir, "drm", f"renderD{minor}"), exist_ok=True)
    _write(os.path.join(pcidir, "pp_dpm_sclk"), f"0: 500Mhz \n{sclk_line}\n2: 2400Mhz \n")
    _write(os.path.join(pcidir, "pp_dpm_mclk"), "0: 2000Mhz *\n")
    _write(os.path.join(pcidir, "pp_dpm_fclk"), "0: 1250Mhz *\n")
    _write(os.path.join(pcidir, "gpu_busy_percent"), f"{busy}\n")
    _write(os.path.join(pcidir, "mem_busy_percent"), "65\n")
    _write(os.path.join(pcidir, "unique_id"), uid + "\n")
    _write(os.path.join(pcidir, "vbios_version"), "113-M355-01-1K1-020F\n")
    _write(os.path.join(pcidir, "current_compute_partition"), "SPX\n")
    _write(os.path.join(pcidir, "current_memory_partition"), "NPS1\n")
    _write(os.path.join(pcidir, "mem_info_vram_total"), "309220868096\n")
    _write(os.path.join(pcidir, "mem_info_vram_used"), "297766912\n")
    _write(os.path.join(pcidir, "numa_node"), "1\n")
    _write(os.path.join(pcidir, "ras", "gpu_vram_bad_pages"), "0x0000 : 0x0000 : R\n")
    hw = os.path.join(pcidir, "hwmon", "hwmon3")
    _write(os.path.join(hw, "power1_average"), "1221000000\n")
    _write(os.path.join(hw, "power1_cap"), "1400000000\n")
    _write(os.path.join(hw, "temp3_input"), "51000\n")
    return pcidir


@pytest.fixture
def fake_box(tmp_path, monkeypatch):
    root = str(tmp_path)
    _card(root, "card0", "0000:75:00.0", 128, "3b0d115c3e11a5b8", 0, "S: 95Mhz *")          # somebody else's GPU, idle
    _card(root, "card56", "0000:a4:00.0", 184, "a3873f7f14b22f2f", 100, "1: 2394Mhz *")      # the leased one
    os.makedirs(os.path.join(root, "drm", "card0-DP-1"), exist_ok=True)                      # connectors are not cards
    os.makedirs(os.path.join(root, "dri"), exist_ok=True)
    _write(os.path.join(root, "dri", "renderD184"), "")                                       # only this device node exists
    node = os.path.join(root, "kfd", "topology", "nodes", "3")
    _write(os.path.join(node, "properties"), "simd_count 1024\ncu_count 256\nnum_xcc 8\nmax_engine_clk_fcompute 2400\ndrm_render_minor 184\n")
    _write(os.path.join(node, "gpu_id"), "23660\n")
    other = os.path.join(root, "kfd", "topology", "nodes", "2")
    _write(os.path.join(other, "properties"), "simd_count 1024\ncu_count 256\ndrm_render_minor 128\n")
    _write(os.path.join(other, "gpu_id"), "36622\n")
    _write(os.path.join(root, "kfd", "proc", "4242", "vram_36622"), "882561024\n")          # a tenant on the OTHER GPU
    smi = os.path.join(root, "rocm-smi")
    counter = os.path.join(root, "calls")
    _write(smi, f"""#!{sys.executable}
import json, os
n = int(open({counter!r}).read()) if os.path.exists({counter!r}) else 0
open({counter!r}, "w").write(str(n + 1))
print("WARNING: some banner line")
print(json.dumps({{"card0": {{"accumulation_counter (Count)": str(1000 + 1000 * n), "ppt_residency_acc (Count)": str(10 + 60 * n),
      "hbm_thm_residency_acc (Count)": "0", "gfx_activity_acc (%)": str(93000 * n), "mem_activity_acc (%)": str(60000 * n),
      "energy_accumulator (15.259uJ (2^-16))": str(int(1e9 + n * 7.6e7)), "current_uclk (MHz)": "2000", "current_gfxclk (MHz)": "2394",
      "temperature_mem (C)": "50", "throttle_status": "N/A", "xcp_stats.gfx_below_host_limit_acc (Count)": "['N/A', 'N/A']"}}}}))
""")
    os.chmod(smi, os.stat(smi).st_mode | stat.S_IEXEC)
    monkeypatch.setattr(bt, "DRM", os.path.join(root, "drm"))
    monkeypatch.setattr(bt, "KFD", os.path.join(root, "kfd"))
    monkeypatch.setattr(bt, "DEV_DRI", os.path.join(root, "dri"))
    monkeypatch.setattr(bt, "ROCM_SMI", smi)
    monkeypatch.setattr(bt, "AMD_SMI", os.path.join(root, "no-amd-smi"))
    monkeypatch.setattr(bt, "DEFAULT_PCI", None)
    return root


def test_the_leased_card_is_the_one_whose_render_node_opens(fake_box):
    assert [os.path.basename(os.path.dirname(d)) for d in bt.cards()] == ["card0", "card56"]
    assert os.path.basename(os.path.realpath(bt.my_card())) == "0000:a4:00.0"
    assert os.path.basename(os.path.realpath(bt.my_card("0000:75:00.0"))) == "0000:75:00.0"       # an explicit bus id wins
    bt.DEFAULT_PCI = "0000:75:00.0"                                                               # ... so does the HIP runtime's answer
    assert os.path.basename(os.path.realpath(bt.my_card())) == "0000:75:00.0"
    bt.DEFAULT_PCI = None
    snap = bt.snapshot(tools=False)
    c = bt.condensed(snap)
    assert c["unique_id"] == "a3873f7f14b22f2f" and c["pci"] == "0000:a4:00.0" and c["render_node_usable"]
    assert (c["sclk"], c["sclk_max"], c["mclk"], c["fclk"]) == (2394, 2400, 2000, 1250)
    assert c["power_w"] == 1221.0 and c["power_cap_w"] == 1400 and c["hbm_temp_c"] == 51.0
    assert c["compute_partition"] == "SPX" and c["memory_partition"] == "NPS1" and c["ras_bad_pages"] == 1   # one retired-page record in the fixture
    assert c["num_cu"] == 256 and c["num_xcc"] == 8 and c["max_engine_clk"] == 2400
    assert c["vram_of_kfd_processes_on_my_gpu"] == 0 and c["vram_processes_on_other_gpus_of_the_node"] == 882561024
    assert c["node_gpus_visible"] == 2 and c["numa_node"] == 1
    json.dumps(snap)   # everything is JSON-able


def test_sampler_and_accumulator_deltas(fake_box):
    a = bt.metrics()
    with bt.Sampler(period_s=0.002) as s:
        time.sleep(0.05)
    b = bt.metrics()
    summ = s.summary()
    assert summ["samples"] >= 3 and summ["sclk"] == [2394, 2394, 2394] and summ["gpu_busy"] == [100, 100, 100] and summ["power_w"][1] == 1221.0
    d = bt.metrics_delta(a, b)
    assert d["firmware_samples"] == 1000 and d["ppt_residency_frac"] == 0.06 and d["hbm_thm_residency_frac"] == 0.0
    assert d["gfx_activity_mean_pct"] == 93.0 and d["mem_activity_mean_pct"] == 60.0 and d["current_uclk_end"] == 2000
    assert d["mean_socket_power_w"] > 0 and d["gfx_below_host_limit_acc_delta"] == []           # 'N/A' entries are skipped, not summed
    assert "error" in bt.metrics_delta({"error": "x"}, b)


def test_nothing_raises_where_nothing_exists(tmp_path, monkeypatch):
    monkeypatch.setattr(bt, "DRM", str(tmp_path / "nothing"))
    monkeypatch.setattr(bt, "KFD", str(tmp_path / "nothing"))
    monkeypatch.setattr(bt, "ROCM_SMI", str(tmp_path / "nothing"))
    monkeypatch.setattr(bt, "AMD_SMI", str(tmp_path / "nothing"))
    assert bt.cards() == [] and bt.my_card() is None
    snap = bt.snapshot()
    assert "error" in snap and "error" in bt.condensed(snap)
    assert "error" in bt.metrics() and "error" in bt.metrics_delta(bt.metrics(), bt.metrics())
    s = bt.Sampler()
    s.start()
    s.stop()
    assert s.summary()["samples"] == 0
    assert bt.fast_read(str(tmp_path / "nothing"))["sclk"] is None


def test_smi_tools_are_never_started_under_a_profiler_and_never_through_env(tmp_path, monkeypatch):
    """`rocm-smi` is a `#!/usr/bin/env python3` script.  Under `rocprofv3 --pmc -- python3 bench.py` a child
    would inherit the profiler's preload, initialise the GPU inside `env`, and `env` would exec python3 -- the exec that takes a
    box of this pool down.  `_tool` refuses under a profiler, strips the profiler's variables otherwise, and starts a script with
    this interpreter on its real path (no `env` hop)."""
    import box_telemetry as bt

    script = tmp_path / "fake-smi"
    script.write_text("#!/usr/bin/env python3\nimport json, os, sys\n"
                      "print(json.dumps({'argv0': sys.argv[0], 'exe': sys.executable, 'args': sys.argv[1:],\n"
                      "                  'leaked': sorted(k for k in os.environ if k.startswith(('ROCP', 'HSA_TOOLS')))}))\n")
    script.chmod(0o755)
    link = tmp_path / "smi-link"
    link.symlink_to(script)
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    monkeypatch.setenv("HSA_TOOLS_LIB", "librocprofiler-sdk-tool.so")      # what a profiler leaves for its children
    seen = {}
    real_run = subprocess.run

    def spy(argv, **kw):
        seen["argv"], seen["env"] = list(argv), kw.get("env")
        return real_run(argv, **kw)

    monkeypatch.setattr(bt.subprocess, "run", spy)
    rc, out, err = bt._tool([str(link), "--showmetrics"])
    assert rc == 0, err
    rec = json.loads(out)
    assert seen["argv"][:2] == [sys.executable, str(script)] and rec["args"] == ["--showmetrics"]    # interpreter + real path: no env, no shebang
    assert rec["leaked"] == [] and "HSA_TOOLS_LIB" not in seen["env"]
    for var, val in (("ROCPROFILER_LIBRARY", "x"), ("ROCP_TOOL_LIBRARIES", "x"), ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")):
        monkeypatch.setenv(var, val)
        seen.clear()
        assert bt.under_profiler()
        rc, out, err = bt._tool([str(link), "--showmetrics"])
        assert rc is None and out == "" and "profiler" in err and not seen       # nothing was started
        assert "error" in bt._json_tool([str(link)]) and "error" in (bt.metrics() if os.path.exists(bt.ROCM_SMI) else {"error": 1})
        monkeypatch.delenv(var)
    assert not bt.under_profiler()
