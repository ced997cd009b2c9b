"""The GPUDistributor / MultiGPUProcessor mirror (gpu_distributor.py) against plans produced by the reference's own class
(tests/golden/gpu_distributor_reference.json, oracle/gen_golden.py gpu_distributor_logic()), and the thread-pool driver with a
stand-in process_func (no GPU needed: the executors and the bookkeeping are host code)."""
import json
import threading
from pathlib import Path

import numpy as np
import pytest

from framewright_amd import gpu_distributor as G

GOLD = json.loads((Path(__file__).parent / "golden" / "gpu_distributor_reference.json").read_text())


def _make(table, stats, strategy=G.DistributionStrategy.LOAD_BALANCED):
    gd = G.GPUDistributor(strategy)
    gd._devices = [G.DeviceInfo(index=i, name=f"gpu{i}", vendor=G.GPUVendor.AMD, total_memory_mb=t, free_memory_mb=f) for i, t, f in table]
    for dev in gd._devices:
        st = G.GPUStats(device_id=dev.index, vendor=dev.vendor, name=dev.name, total_memory_mb=dev.total_memory_mb)
        if str(dev.index) in stats:
            cnt, avg = stats[str(dev.index)]
            st.frames_processed, st.avg_time_per_frame, st.total_time_seconds = int(cnt), avg, cnt * avg
        gd._stats[dev.index] = st
    return gd


def test_every_planner_matches_the_reference():
    assert len(GOLD["cases"]) >= 70
    for c in GOLD["cases"]:
        for strat in G.DistributionStrategy:
            gd = _make(c["devices"], c["stats"])
            want = c["plans"][strat.value]
            if "error" in want:
                with pytest.raises(Exception) as ei:
                    gd.distribute_frames(c["n"], strat)
                assert type(ei.value).__name__ == want["error"]
                continue
            plan = gd.distribute_frames(c["n"], strat)
            assert {str(k): v for k, v in plan.gpu_workloads.items()} == want, (c["table"], c["timing"], c["n"], strat)
            assert plan.total_frames == c["n"]
            assert sorted(plan.frame_assignments) == list(range(c["n"]))
            for dev, idxs in plan.gpu_workloads.items():
                assert all(plan.frame_assignments[i] == dev for i in idxs)
        gd = _make(c["devices"], c["stats"])
        if "error" in c["optimal"]:
            with pytest.raises(Exception):
                gd.get_optimal_distribution(c["n"])
        else:
            assert {str(k): v for k, v in gd.get_optimal_distribution(c["n"]).gpu_workloads.items()} == c["optimal"]


def test_unhealthy_devices_are_left_out_and_restored():
    gd = _make([(i, 1000, 900) for i in range(3)], {}, G.DistributionStrategy.ROUND_ROBIN)
    gd.mark_device_unhealthy(1)
    assert {str(k): v for k, v in gd.distribute_frames(6).gpu_workloads.items()} == GOLD["unhealthy_rr"]
    gd.mark_device_healthy(1)
    assert sorted(gd.distribute_frames(6).gpu_workloads) == [0, 1, 2]
    gd._stats[0].update_timing(0.5)
    gd.reset_stats()
    assert gd._stats[0].frames_processed == 0 and gd.get_all_stats()[0].is_healthy


def test_process_frames_runs_the_b4_contract_on_per_gpu_pools():
    """process_func(frame, device_id) per frame on that device's executor, results in frame order, failures reported per frame,
    stats updated - with stand-in backends (the real one needs a GPU: tests/test_boundaries.py)."""
    mp = G.MultiGPUProcessor(G.DistributionStrategy.ROUND_ROBIN, max_workers_per_gpu=2)
    mp.distributor._devices = [G.DeviceInfo(index=i, name=f"gpu{i}", vendor=G.GPUVendor.AMD, total_memory_mb=294912, free_memory_mb=290000)
                               for i in range(4)]
    for d in mp.distributor._devices:
        mp.distributor._stats[d.index] = G.GPUStats(device_id=d.index, vendor=d.vendor, name=d.name)

    class Stub:
        def cleanup(self):
            pass

        def get_memory_info(self):
            return {"free_mb": 1000}
    from concurrent.futures import ThreadPoolExecutor
    for d in mp.distributor._devices:
        mp._backends[d.index] = Stub()
        mp._executors[d.index] = ThreadPoolExecutor(max_workers=2, thread_name_prefix=f"gpu_{d.index}_")
    mp._initialized = True
    seen, lock = [], threading.Lock()

    def fn(frame, device_id):
        with lock:
            seen.append((int(frame[0, 0, 0]), device_id, threading.current_thread().name))
        if frame[0, 0, 0] == 5:
            raise ValueError("boom")
        return frame * 2

    frames = [np.full((2, 2, 3), i, np.uint8) for i in range(10)]
    done = []
    res = mp.process_frames(frames, fn, callback=done.append)
    assert [r.frame_index for r in res] == list(range(10)) and len(done) == 10
    plan = mp.distributor.distribute_frames(10, G.DistributionStrategy.LOAD_BALANCED)
    for r in res:
        assert r.device_id == plan.frame_assignments[r.frame_index]
        if r.frame_index == 5:
            assert not r.success and r.error == "boom" and r.output is None
        else:
            assert r.success and np.array_equal(r.output, frames[r.frame_index] * 2)
    assert all(name.startswith(f"gpu_{dev}_") for _, dev, name in seen)
    st = mp.get_stats()
    assert sum(s.frames_processed for s in st.values()) == 9 and sum(s.errors for s in st.values()) == 1
    assert mp.process_batch(np.zeros((2, 2, 2, 3)), lambda b, d: b + d, device_id=3).max() == 3
    assert mp.get_available_gpus() == [0, 1, 2, 3] and mp.get_gpu_count() == 4
    mp.cleanup()
    assert mp.get_gpu_count() == 0


def test_from_config_maps_the_reference_config_fields():
    class Cfg:
        enable_multi_gpu, gpu_ids, gpu_load_balance_strategy, workers_per_gpu = True, None, "vram_aware", 2
    mp = G.MultiGPUProcessor.from_config(Cfg())
    assert mp.distributor.strategy is G.DistributionStrategy.MEMORY_AWARE and mp.max_workers_per_gpu == 2
    assert G.detect_multi_gpu_support()["gpu_count"] >= 0
