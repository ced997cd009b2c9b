"""MultiGPUDistributor mirror (framewright_amd/distributor.py): the four assignment strategies against vectors the
reference's own `_assign_frames` produced (tests/golden/assign_frames.json, oracle/gen_golden.py assign_frames_logic), and
the worker / retry behaviour with a fake GPU manager (no GPU needed: process_fn is a plain callable)."""
import json
import threading
from pathlib import Path

import pytest

from framewright_amd import distributor as D

GOLD = json.loads((Path(__file__).parent / "golden" / "assign_frames.json").read_text())


def test_assign_frames_matches_reference_for_every_strategy():
    assert len(GOLD) == 240
    for case in GOLD:
        gpus = [D.GPUInfo(**g) for g in case["gpus"]]
        plan = D.assign_frames(list(range(case["n_frames"])), gpus, D.LoadBalanceStrategy(case["strategy"]))
        assert {str(k): v for k, v in plan.items()} == case["plan"], (case["strategy"], case["n_frames"], case["gpus"])
        assert list(map(str, plan.keys())) == list(case["plan"].keys())           # same GPU order in the dict


class FakeManager:
    def __init__(self, n):
        self.gpus = [D.GPUInfo(id=i, name=f"fake{i}", total_vram_mb=1000, free_vram_mb=1000 - 100 * i, utilization_pct=10.0 * i)
                     for i in range(n)]

    def get_healthy_gpus(self):
        return list(self.gpus)

    get_all_gpu_info = get_healthy_gpus


def test_distribute_frames_retries_on_another_gpu_and_reports(tmp_path):
    frames = [tmp_path / f"f{i:03d}.png" for i in range(23)]
    seen, lock = [], threading.Lock()

    def process_fn(path, outdir, gpu_id):
        with lock:
            seen.append((path.name, gpu_id))
        if path.name == "f004.png" and gpu_id != 2:
            return path, False, "device busy"          # fails everywhere but on GPU 2
        if path.name == "f007.png":
            return path, False, "corrupt frame"        # fails on every GPU it is tried on
        if path.name == "f009.png":
            raise RuntimeError("boom")                 # an exception fails the frame without a retry
        return outdir / path.name, True, None

    prog = []
    d = D.MultiGPUDistributor(FakeManager(3), strategy=D.LoadBalanceStrategy.ROUND_ROBIN, workers_per_gpu=2)
    res = d.distribute_frames(frames, process_fn, tmp_path / "out", lambda p, m: prog.append(p))
    assert res.total_frames == 21 and set(res.errors) == {str(tmp_path / "f007.png"), str(tmp_path / "f009.png")}
    assert res.errors[str(tmp_path / "f007.png")] == "corrupt frame" and res.errors[str(tmp_path / "f009.png")] == "boom"
    assert (tmp_path / "out" / "f004.png") in res.frames_per_gpu[2] and (tmp_path / "f004.png") in res.retried_frames
    assert sorted(g for n, g in seen if n == "f007.png") == [0, 1, 2]            # three attempts, three different GPUs
    assert [g for n, g in seen if n == "f009.png"] == [0]
    assert len(prog) == 23 and prog[-1] == 1.0 and d.get_result() is res
    assert abs(res.success_rate - 100 * 21 / 23) < 1e-9 and "across 3 GPUs" in res.summary()
    assert (tmp_path / "out").is_dir()


def test_no_gpus_and_empty_input(tmp_path):
    d = D.MultiGPUDistributor(FakeManager(0))
    res = d.distribute_frames([tmp_path / "a.png"], lambda *a: (a[0], True, None), tmp_path)
    assert res.errors == {str(tmp_path / "a.png"): "No GPUs available"}
    assert D.MultiGPUDistributor(FakeManager(2)).distribute_frames([], None, tmp_path).total_frames == 0
    g = D.GPUInfo(id=0, name="x", total_vram_mb=1000, free_vram_mb=250, utilization_pct=50.0, temperature_c=95.0)
    assert g.used_vram_mb == 750 and g.vram_usage_pct == 75.0 and not g.is_healthy
    assert abs(g.effective_capacity - (0.25 * 0.7 + 0.5 * 0.3)) < 1e-12
