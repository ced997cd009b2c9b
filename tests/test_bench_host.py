"""bench.py's contract, as far as it can be checked without a GPU: the launcher check, the byte models behind the `rife` / `tap`
roofline lines, and the measured-traffic files a line may quote (only for the build they were taken on)."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("fw_bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_without_the_launcher_is_refused_before_any_gpu_call():
    """`python bench.py --gpus N` alone used to run one rank and report n_gpus: 1: it must exit non-zero and say how to
    launch - the driver starts N > 1 under torch.distributed.run, one rank per GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" in (r.stderr + r.stdout) and "--nproc-per-node 2" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")          # no JSON line was printed


def test_byte_models_scale_with_the_frame():
    b = _bench()
    small, full = b.nafnet_design_bytes(270, 480), b.nafnet_design_bytes(1080, 1920)
    assert 0 < small < full and 14 < full / small < 18            # 16 x the pixels (padding to multiples of 16 aside)
    assert 35e9 < full < 45e9                                       # 40.5 GB per 1080p forward in this dataflow
    s2, f2 = b.ifnet_design_bytes(270, 480), b.ifnet_design_bytes(1080, 1920)
    assert 0 < s2 < f2 and 3e9 < f2 < 6e9


def test_traffic_profiles_are_tagged_with_a_build_digest():
    """A bench line quotes measured HBM traffic only when the file's digest names the library it runs: every traffic file of the
    latest round carries one, the dtype, and per-launch / per-forward byte counts in a sane range."""
    b = _bench()
    sr = b.newest_profile("traffic.json")
    assert sr is not None and sr.name >= "r03_traffic.json"
    t = json.loads(sr.read_text())
    assert len(t["lib_digest"]) == 16 and t["dtype"] in ("f16", "bf16")
    assert 150 < t["hbm_gb_per_frame"] < 300 and abs(t["frames"] - round(t["frames"])) < 1e-6
    for name in ("traffic_tap.json", "traffic_rife.json"):
        f = b.newest_profile(name)
        assert f is not None, name
        tj = json.loads(f.read_text())
        assert len(tj["lib_digest"]) == 16 and tj["hbm_bytes_per_forward"] > 1e9
    assert b.newest_profile("no_such_file.json") is None


def test_power_sampler_never_takes_the_bench_down(tmp_path, monkeypatch):
    """The `energy` object of a bench line comes from rocm-smi sampled in a child process; without the tool, without a GPU or with a run too
    short for two samples the sampler answers None, and with samples it reports the medians of the region under load."""
    b = _bench()
    s = b.PowerSampler(0)
    assert s.stop() is None or {"power_w", "sclk_mhz", "samples"} <= set(s.stop() or {"power_w": 0, "sclk_mhz": 0, "samples": 0})
    # a stand-in rocm-smi: three samples under load, one idle
    fake = tmp_path / "rocm-smi"
    state = tmp_path / "n"
    fake.write_text("#!/bin/bash\nn=$(cat %s 2>/dev/null || echo 0); echo $((n + 1)) > %s\n"
                    "if [ $n -lt 3 ]; then p=1399.0; c=1750; else p=400.0; c=2400; fi\n"
                    "echo \"GPU[0]\t\t: Current Socket Graphics Package Power (W): $p\"\necho \"GPU[0]\t\t: sclk clock level: 1: (${c}Mhz)\"\n" % (state, state))
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    s = b.PowerSampler(0)
    import time
    for _ in range(100):
        if state.exists() and int(state.read_text() or 0) >= 4:
            break
        time.sleep(0.1)
    r = s.stop()
    assert r is not None and r["power_w"] == 1399.0 and r["sclk_mhz"] == 1750 and r["samples"] >= 3
