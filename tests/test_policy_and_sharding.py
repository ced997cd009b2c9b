"""Host bookkeeping pinned on outputs of the reference's own functions (tests/golden/host_logic.json, written by
oracle/gen_golden.py), and the multi-GPU sharding on 2 gloo ranks (CPU)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from framewright_amd import policy as P
from framewright_amd import sharding as S
from framewright_amd import tap_denoise as T
from oracle import tap_ref


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.loads((golden_dir / "host_logic.json").read_text())


def test_tile_policy_matches_reference(golden):
    assert len(golden["tile_policy"]) == 160
    for c in golden["tile_policy"]:
        t = P.calculate_optimal_tile_size(tuple(c["res"]), c["scale"], c["vram"], c["model"])
        assert t == c["tile"], c
        assert P.get_adaptive_tile_sequence(tuple(c["res"]), c["scale"], t) == c["seq"], c
    for c in golden["tile_sequences"]:
        assert P.get_adaptive_tile_sequence(tuple(c["res"]), c["scale"], c["start"], c["min"]) == c["seq"], c
    # the two values the reference's own tests pin (tests/test_utils_gpu.py:130-210): plenty of VRAM -> no tiling
    assert P.calculate_optimal_tile_size((1920, 1080), 4, 288 * 1024) == 0


def test_interpolation_strategy_matches_reference(golden):
    for want in golden["interp_factors"]:
        got = P.calculate_interpolation_factor(want["source_fps"], want["target_fps"])
        assert got == want
    assert [P.interpolation_exponent(f) for f in (1.5, 2, 2.5, 4, 8, 9)] == [1, 1, 2, 2, 3, 4]
    assert P.interp_fps_for_target(24, 60) == 96 and P.interp_fps_for_target(24, 30) == 48 and P.interp_fps_for_target(10, 240) == 80


def test_decimation_matches_reference_loop(golden):
    for c in golden["decimation"]:
        assert P.decimation_indices(c["n"], c["interp_fps"], c["target_fps"]) == c["keep"], c


def test_histogram_scene_detection_matches_reference(golden):
    for c in golden["histogram_scene"]:
        a, b = np.array(c["a"], dtype=np.uint8), np.array(c["b"], dtype=np.uint8)
        assert P.scene_change_by_histogram(a, b, c["threshold"]) == c["scene_change"]
    assert any(c["scene_change"] for c in golden["histogram_scene"]) and not all(c["scene_change"] for c in golden["histogram_scene"])


def test_round_robin_matches_reference_planner(golden):
    for c in golden["round_robin"]:
        assert "error" not in c, c
        got = S.round_robin_assignment(c["n"], c["devices"])
        assert {str(k): v for k, v in got.items()} == c["workloads"]
    assert S.block_partition(304, 8) == [(38 * r, 38 * (r + 1)) for r in range(8)]      # BASELINE config 4
    assert S.block_partition(5, 2) == [(0, 2), (2, 5)]


# ---- 2-rank gloo -----------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_denoise(f):
    return (f.astype(np.float32) * 0.9 + 7).astype(np.uint8)


def _combine(window, start, center, n):
    s, e, ws = tap_ref.temporal_weights(n, center, 5)
    assert s == start and e - s == len(window)
    return tap_ref.temporal_average(window, ws)


def _worker(rank, world, port, frames, q, as_tensors=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if as_tensors:      # the device-resident form: tensors in, tensors (halos included) through, no numpy staging
            import torch
            tf = [torch.from_numpy(f) for f in frames]
            den = S.sharded_temporal_denoise(tf, 2, lambda t: torch.from_numpy(_fake_denoise(t.numpy())),
                                             lambda w, s, c, n: _combine([t.numpy() for t in w], s, c, n))
            mid = S.sharded_pairs(tf, lambda a, b: ((a.to(torch.int32) + b.to(torch.int32)) // 2).to(torch.uint8))
            mid = {k: v.numpy() for k, v in mid.items()}
        else:
            den = S.sharded_temporal_denoise(frames, 2, _fake_denoise, _combine)
            mid = S.sharded_pairs(frames, lambda a, b: ((a.astype(np.uint16) + b) // 2).astype(np.uint8))
        up = S.sharded_upscale(frames, lambda f: np.repeat(np.repeat(f, 2, 0), 2, 1))
        q.put((rank, den, {k: v.shape for k, v in up.items()}, mid))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_block_partition_with_fewer_frames_than_ranks():
    # ranks without frames sit at the end, and every rank derives the same table
    assert S.block_partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert S.block_partition(3, 4, 2) == [(0, 3), (3, 3), (3, 3), (3, 3)]
    assert S.block_partition(0, 3) == [(0, 0)] * 3
    assert S.block_partition(9, 4, 2) == [(0, 2), (2, 4), (4, 6), (6, 9)]
    for n in range(0, 12):
        for world in (1, 2, 3, 4, 8):
            for r in (1, 2, 3):
                plans = [S.temporal_halo_plan(n, world, k, r) for k in range(world)]
                for k in range(world - 1):   # both ends of every link agree on what crosses it
                    assert plans[k][2][0] == plans[k + 1][1][1] and plans[k][2][1] == plans[k + 1][1][0]
                assert plans[0][1] == (0, 0) and plans[-1][2] == (0, 0)


@pytest.mark.parametrize("n_frames,world,as_tensors", [
    (7, 2, False), (10, 2, False), (10, 2, True), (2, 4, False), (3, 4, True), (5, 3, False),
    # the node the driver's scaling run uses: eight ranks on the clip lengths of SURVEY section 8(d) (304 = 8 x 38 frames for the
    # temporal denoise, 600 for the chain) and on a clip shorter than the world (ranks without frames post nothing and exit)
    (304, 8, True), (600, 8, False), (5, 8, True)])
def test_two_rank_sharding_equals_single_process(n_frames, world, as_tensors):
    rng = np.random.default_rng(n_frames)
    frames = [rng.integers(0, 256, size=(12, 16, 3), dtype=np.uint8) for _ in range(n_frames)]
    want_den = {i: _combine([_fake_denoise(f) for f in frames[max(0, i - 2):i + 3]], max(0, i - 2), i, n_frames)
                for i in range(n_frames)}
    want_mid = {i: ((frames[i].astype(np.uint16) + frames[i + 1]) // 2).astype(np.uint8) for i in range(n_frames - 1)}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, q, as_tensors)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    den, ups, mid = {}, {}, {}
    for rank, d, u, m in got:
        assert set(d) == set(range(*S.block_partition(n_frames, world, 2)[rank]))   # block partition, no overlap
        assert set(u) == set(S.round_robin_assignment(n_frames, world)[rank])       # round-robin for SR
        den.update(d)
        ups.update(u)
        mid.update(m)
    assert set(den) == set(range(n_frames)) and set(ups) == set(range(n_frames)) and set(mid) == set(range(n_frames - 1))
    for i in range(n_frames):
        np.testing.assert_array_equal(den[i], want_den[i])
        assert ups[i] == (24, 32, 3)
    for i in range(n_frames - 1):
        np.testing.assert_array_equal(mid[i], want_mid[i])


def test_single_process_sharding_is_identity_partition():
    frames = [np.full((4, 4, 3), i, np.uint8) for i in range(5)]
    out = S.sharded_upscale(frames, lambda f: f)
    assert sorted(out) == [0, 1, 2, 3, 4]
    assert T.temporal_window(5, 2, 5)[:2] == (0, 5)
