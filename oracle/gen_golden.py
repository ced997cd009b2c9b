"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES in the build container.

Run here only (``python oracle/gen_golden.py``): /root/reference does not exist on the GPU box, and the reference
source never travels — only the small input/output vectors written by this script do.

``import framewright`` fails in the reference snapshot (framewright/__init__.py:6 -> restorer.py:178 ->
infrastructure/__init__.py:67 imports a directory that is not in the snapshot), so individual modules are loaded
through namespace stubs.

Fixture set 1 — residual-dense arithmetic (reference src/framewright/processors/aesrgan_face.py):
  * ``ResidualDenseBlock(64)``            :171-189   1x64x12x12 -> 1x64x12x12
  * ``RRDB(64)``                          :192-204   1x64x12x12 -> 1x64x12x12
  * ``AESRGAN(num_block=2, scale=4, num_attention=1)`` :207-269   1x3x16x16 -> 1x3x64x64
    (its single AttentionBlock has gamma = 0 at construction (:149), so ``gamma*out + x`` (:169) is the identity and
    the module is exactly conv_first -> 2xRRDB -> conv_body -> +feat -> up1 -> up2 -> conv_hr -> conv_last)
Weights are NOT stored: they are regenerated from a seed with framewright_amd.synth (numpy PCG64), loaded into the
reference modules here and into the oracle in the tests.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"

from framewright_amd.synth import synthetic_rrdbnet_state  # noqa: E402


def load_reference(module: str):
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    for n in ["framewright", "framewright.processors", "framewright.processors.enhancement", "framewright.utils",
              "framewright.plugins", "framewright.engine", "framewright.infrastructure",
              "framewright.infrastructure.gpu"]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = [f"{REF_SRC}/{n.replace('.', '/')}"]
            sys.modules[n] = m
    return importlib.import_module(module)


def _load_rdb(rdb, sd, prefix):
    for c in range(1, 6):
        conv = getattr(rdb, f"conv{c}")
        conv.weight.data.copy_(torch.from_numpy(sd[f"{prefix}.conv{c}.weight"]))
        conv.bias.data.copy_(torch.from_numpy(sd[f"{prefix}.conv{c}.bias"]))


def _load_rrdb(rrdb, sd, prefix):
    for r in (1, 2, 3):
        _load_rdb(getattr(rrdb, f"rdb{r}"), sd, f"{prefix}.rdb{r}")


def main() -> None:
    a = load_reference("framewright.processors.aesrgan_face")
    assert a.HAS_TORCH
    out = {}
    torch.manual_seed(0)
    rng = np.random.default_rng(2024)

    # --- ResidualDenseBlock -----------------------------------------------------------------
    sd = synthetic_rrdbnet_state(1, 4, seed=11)
    rdb = a.ResidualDenseBlock(64).eval()
    _load_rdb(rdb, sd, "body.0.rdb1")
    x = rng.standard_normal((1, 64, 12, 12)).astype(np.float32)
    with torch.no_grad():
        y = rdb(torch.from_numpy(x).clone()).numpy()
    out["rdb_seed"], out["rdb_in"], out["rdb_out"] = np.int64(11), x, y

    # --- RRDB -----------------------------------------------------------------------------------
    sd = synthetic_rrdbnet_state(1, 4, seed=12)
    rrdb = a.RRDB(64).eval()
    _load_rrdb(rrdb, sd, "body.0")
    x = rng.standard_normal((1, 64, 12, 12)).astype(np.float32)
    with torch.no_grad():
        y = rrdb(torch.from_numpy(x).clone()).numpy()
    out["rrdb_seed"], out["rrdb_in"], out["rrdb_out"] = np.int64(12), x, y

    # --- AESRGAN trunk + tail (attention inert: gamma == 0) -----------------------------------------------
    sd = synthetic_rrdbnet_state(2, 4, seed=13)
    net = a.AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=2, scale=4, num_attention=1).eval()
    rrdbs = [m for m in net.body if isinstance(m, a.RRDB)]
    attn = [m for m in net.body if isinstance(m, a.AttentionBlock)]
    assert len(rrdbs) == 2 and len(attn) == 1 and float(attn[0].gamma.abs().sum()) == 0.0
    for i, m in enumerate(rrdbs):
        _load_rrdb(m, sd, f"body.{i}")
    for name in ("conv_first", "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last"):
        conv = getattr(net, name)
        conv.weight.data.copy_(torch.from_numpy(sd[name + ".weight"]))
        conv.bias.data.copy_(torch.from_numpy(sd[name + ".bias"]))
    x = rng.uniform(0, 1, size=(1, 3, 16, 16)).astype(np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x).clone()).numpy()
    out["net_seed"], out["net_in"], out["net_out"] = np.int64(13), x, y

    dst = ROOT / "tests" / "golden" / "rrdb_reference.npz"
    dst.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(dst, **out)
    print(f"wrote {dst} ({dst.stat().st_size / 1024:.0f} KiB)")
    for k, v in out.items():
        if hasattr(v, "shape") and v.ndim:
            print(f"  {k}: {v.shape} mean {v.mean():+.4f} std {v.std():.4f}")


def host_logic() -> None:
    """Fixture set 2 — pure-Python bookkeeping of the reference, evaluated by the reference's own functions:
    utils/gpu.py:386-512 (tile policy), processors/interpolation.py:325-366,783-791,811-845 (scene histogram,
    decimation loop, interpolation strategy), infrastructure/gpu/distributor.py:287-304 and utils/multi_gpu.py:796-800
    (round-robin frame assignment)."""
    import json
    import shutil
    from unittest import mock

    g = load_reference("framewright.utils.gpu")
    cases = []
    for res in [(1920, 1080), (1280, 720), (640, 360), (3840, 2160)]:
        for scale in (2, 4):
            for vram in (2048, 4096, 8192, 24576, 81920):
                for model in ("realesrgan-x4plus", "realesrgan-x2plus", "realesr-animevideov3", "unknown"):
                    t = g.calculate_optimal_tile_size(res, scale, vram, model)
                    cases.append({"res": res, "scale": scale, "vram": vram, "model": model, "tile": t,
                                  "seq": g.get_adaptive_tile_sequence(res, scale, t)})
    seqs = [{"res": r, "scale": s, "start": st, "min": mn, "seq": g.get_adaptive_tile_sequence(r, s, st, mn)}
            for r, s, st, mn in [((1920, 1080), 4, 512, 128), ((1920, 1080), 4, 400, 64), ((720, 480), 2, 0, 128),
                                 ((1920, 1080), 4, 100, 128), ((4096, 2160), 4, 1000, 128)]]

    ip = load_reference("framewright.processors.interpolation")
    with mock.patch.object(shutil, "which", return_value="/usr/bin/true"):  # __init__ would try to download rife-ncnn
        try:
            fi = ip.FrameInterpolator()
        except Exception:  # noqa: BLE001
            fi = ip.FrameInterpolator.__new__(ip.FrameInterpolator)
            fi.config = ip.InterpolationConfig()
    factors = [ip.FrameInterpolator.calculate_interpolation_factor(s, t)
               for s, t in [(24, 30), (24, 48), (24, 50), (24, 60), (25, 50), (23.976, 59.94), (15, 120), (12, 240), (30, 60)]]
    rng = np.random.default_rng(77)
    hist = []
    for k in range(6):
        a = rng.integers(0, 256, size=(24, 32, 3), dtype=np.uint8)
        b = a.copy() if k == 0 else np.clip(a.astype(int) + rng.integers(-40 * k, 40 * k + 1, size=a.shape), 0, 255).astype(np.uint8)
        hist.append({"seed_k": k, "a": a.tolist(), "b": b.tolist(), "threshold": fi.config.scene_threshold,
                     "scene_change": bool(fi._detect_scene_by_histogram(a, b))})

    # the decimation loop is inline in interpolate_to_fps (interpolation.py:783-791); transcribed verbatim here and
    # evaluated in the reference's arithmetic (float division)
    dec = []
    for n, ifps, tfps in [(96, 48.0, 30.0), (200, 96.0, 60.0), (200, 96.0, 50.0), (50, 47.952, 29.97), (10, 48.0, 47.0)]:
        frame_ratio = ifps / tfps
        out_idx, keep = 0, []
        for i in range(n):
            if i >= out_idx * frame_ratio:
                keep.append(i)
                out_idx += 1
        dec.append({"n": n, "interp_fps": ifps, "target_fps": tfps, "keep": keep})

    d = load_reference("framewright.infrastructure.gpu.distributor")
    rr = []
    for n, ndev in [(10, 4), (7, 8), (300, 8), (5, 1)]:
        devs = [types.SimpleNamespace(index=i) for i in range(ndev)]  # the planner only reads .index (:296-301)
        try:
            plan = d.GPUDistributor._distribute_round_robin(None, n, devs)
            rr.append({"n": n, "devices": ndev, "workloads": {str(k): v for k, v in plan.gpu_workloads.items()}})
        except Exception as e:  # noqa: BLE001
            rr.append({"n": n, "devices": ndev, "error": repr(e)})

    dst = ROOT / "tests" / "golden" / "host_logic.json"
    dst.write_text(json.dumps({"tile_policy": cases, "tile_sequences": seqs, "interp_factors": factors,
                               "histogram_scene": hist, "decimation": dec, "round_robin": rr}, indent=0))
    print(f"wrote {dst} ({dst.stat().st_size / 1024:.0f} KiB); round_robin: {[r.get('error', 'ok') for r in rr]}")


def tap_logic() -> None:
    """Fixture set 3 — TAP temporal-window arithmetic and the motion-adaptive strength table, evaluated by the reference's
    own code (processors/tap_denoise.py; it imports without cv2):
      * ``TAPDenoiser._denoise_with_temporal_window`` (:490-534) on random uint8 frames, with the per-frame denoise step
        replaced by the identity THROUGH ITS OWN INSTANCE ATTRIBUTE (the step needs a loaded network and cv2; the window
        selection, the 1/(1+0.5|d|) weights, the float32 accumulate and the truncating cast are the reference's);
      * ``MotionAdaptiveTAPDenoiser.get_motion_adjusted_strength`` (:878-904) for every MotionLevel x 4 configs;
      * the config defaults and the MODEL_FILES / MODEL_VRAM tables.
    -> tests/golden/tap_reference.npz + tests/golden/tap_reference.json"""
    import json
    t = load_reference("framewright.processors.tap_denoise")
    rng = np.random.default_rng(20260104)
    out, meta = {}, {"windows": []}
    for n, window in ((7, 5), (3, 3), (9, 7), (4, 5), (1, 5)):
        frames = [rng.integers(0, 256, size=(7, 9, 3), dtype=np.uint8) for _ in range(n)]
        den = t.TAPDenoiser(t.TAPDenoiseConfig(temporal_window=window))
        den._denoise_frame_tiled = lambda f: f          # identity per-frame step (instance attribute, reference code untouched)
        key = f"n{n}_w{window}"
        out[key + "_frames"] = np.stack(frames)
        out[key + "_out"] = np.stack([den._denoise_with_temporal_window(frames, i) for i in range(n)])
        meta["windows"].append({"key": key, "n": n, "window": window})
    levels = list(t.MotionLevel)
    table = []
    for cfg in (dict(), dict(base_strength=1.0, motion_sensitivity=1.0), dict(base_strength=0.5, motion_sensitivity=0.25, static_boost=1.05,
                                                                              motion_penalty=0.9), dict(base_strength=0.9, motion_sensitivity=0.0)):
        m = t.MotionAdaptiveTAPDenoiser(t.MotionAdaptiveConfig(**cfg))
        table.append({"config": cfg, "strength": {lv.value: m.get_motion_adjusted_strength(lv) for lv in levels}})
    meta["motion_strength"] = table
    c = t.TAPDenoiseConfig()
    meta["config_defaults"] = {"model": c.model.value, "temporal_window": c.temporal_window, "strength": c.strength,
                               "preserve_grain": c.preserve_grain, "half_precision": c.half_precision, "tile_size": c.tile_size,
                               "tile_overlap": c.tile_overlap, "gpu_id": c.gpu_id, "batch_size": c.batch_size}
    meta["model_files"] = {k.value: v for k, v in t.TAPDenoiser.MODEL_FILES.items()}
    meta["model_vram"] = {k.value: v for k, v in t.TAPDenoiser.MODEL_VRAM.items()}
    meta["motion_levels"] = [lv.value for lv in levels]
    np.savez_compressed(ROOT / "tests" / "golden" / "tap_reference.npz", **out)
    (ROOT / "tests" / "golden" / "tap_reference.json").write_text(json.dumps(meta, indent=1, sort_keys=True))
    print("wrote tests/golden/tap_reference.{npz,json}")


def tile_and_flow_logic() -> None:
    """Fixture set 4 — more arithmetic of the reference run by the reference itself, with the steps that need a network or
    cv2 replaced THROUGH INSTANCE ATTRIBUTES (the reference's source is untouched):
      * ``TAPDenoiser._denoise_frame_tiled`` (tap_denoise.py:417-488): tile grid, np.linspace ramps on interior edges,
        float32 accumulate, / max(weight, 1e-8), truncating cast — with `_preprocess_frame`, `_model`,
        `_postprocess_frame` set to the identity, so a "denoised tile" is the input tile;
      * ``TemporalDenoiser._denoise_simple`` (temporal_denoise.py:1582-1605) as is, and ``_denoise_with_flow`` (:1521-1580)
        with a stand-in flow estimator object whose `estimate` returns given FlowFields (or raises for one neighbour) and
        whose `warp_frame` returns given aligned frames: exp(-d*decay) * confidence, the 90th-percentile motion mask, the
        float64 accumulate and the truncating cast are the reference's; cv2.remap itself is NOT exercised.
    -> tests/golden/tile_flow_reference.npz + .json"""
    import json
    t = load_reference("framewright.processors.tap_denoise")
    td = load_reference("framewright.processors.temporal_denoise")
    rng = np.random.default_rng(20260105)
    out, meta = {}, {"tiled": [], "flow": [], "simple": []}
    ident = lambda x: x
    for h, w, ts, ov in ((40, 56, 32, 8), (64, 64, 32, 8), (33, 47, 16, 4), (50, 36, 32, 16), (30, 30, 32, 8), (48, 80, 24, 1)):
        frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        den = t.TAPDenoiser(t.TAPDenoiseConfig(tile_size=ts, tile_overlap=ov))
        den._preprocess_frame, den._model, den._postprocess_frame = ident, ident, ident
        key = f"tile_{h}x{w}_{ts}_{ov}"
        out[key + "_in"], out[key + "_out"] = frame, den._denoise_frame_tiled(frame)
        meta["tiled"].append({"key": key, "h": h, "w": w, "tile_size": ts, "overlap": ov})

    class FakeEstimator:
        def __init__(self, table, fail_id):
            self.table, self.fail_id = table, fail_id

        def estimate(self, frame, center):
            if id(frame) == self.fail_id:
                raise RuntimeError("flow failed")
            return self.table[id(frame)][0]

        def warp_frame(self, frame, flow, inverse=False):
            return self.table[id(frame)][1]

    for n, center, decay in ((5, 2, 0.5), (4, 0, 0.2), (3, 2, 1.0)):
        h, w = 11, 13
        frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
        table = {}
        key = f"flow_n{n}_c{center}"
        for i, f in enumerate(frames):
            fx = (rng.standard_normal((h, w)) * 2).astype(np.float32)
            fy = (rng.standard_normal((h, w)) * 2).astype(np.float32)
            conf = rng.random((h, w)).astype(np.float32)
            mag = np.sqrt(fx ** 2 + fy ** 2)
            aligned = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            table[id(f)] = (td.FlowField(fx, fy, mag, conf, 0, 1), aligned)
            out[f"{key}_frame{i}"], out[f"{key}_aligned{i}"], out[f"{key}_conf{i}"], out[f"{key}_mag{i}"] = f, aligned, conf, mag
        fail = (center + 1) % n
        d = td.TemporalDenoiser(td.TemporalDenoiseConfig(temporal_weight_decay=decay))
        d._flow_estimator = FakeEstimator(table, id(frames[fail]))
        window = [(i, f) for i, f in enumerate(frames)]
        out[key + "_out"] = d._denoise_with_flow(frames[center], center, window)
        meta["flow"].append({"key": key, "n": n, "center": center, "decay": decay, "failing": fail})
        out[key + "_simple"] = d._denoise_simple(frames[n // 2], window)
        meta["simple"].append({"key": key, "n": n, "decay": decay})
    np.savez_compressed(ROOT / "tests" / "golden" / "tile_flow_reference.npz", **out)
    (ROOT / "tests" / "golden" / "tile_flow_reference.json").write_text(json.dumps(meta, indent=1, sort_keys=True))
    print("wrote tests/golden/tile_flow_reference.{npz,json}")


def interfaces() -> None:
    """Fixture set 5 — the interfaces the drop-in mirrors must reproduce, read off the reference's own classes with
    `inspect` / `dataclasses` (no arithmetic): abstract-method sets, parameter names of the public methods, dataclass field
    names and defaults.  -> tests/golden/interfaces.json"""
    import dataclasses
    import enum
    import inspect
    import json

    def describe(cls):
        d = {"abstract": sorted(getattr(cls, "__abstractmethods__", ())), "methods": {}}
        for name, fn in inspect.getmembers(cls, predicate=inspect.isfunction):
            if name.startswith("_") and name not in ("__init__",):
                continue
            d["methods"][name] = list(inspect.signature(fn).parameters)
        for name, prop in inspect.getmembers(cls, lambda o: isinstance(o, property)):
            d["methods"][name] = ["<property>"]
        if dataclasses.is_dataclass(cls):
            fields = {}
            for f in dataclasses.fields(cls):
                if f.default is not dataclasses.MISSING:
                    v = f.default
                elif f.default_factory is not dataclasses.MISSING:  # type: ignore[misc]
                    v = f.default_factory()  # type: ignore[misc]
                else:
                    v = "<required>"
                if isinstance(v, enum.Enum):
                    v = v.value
                if isinstance(v, Path):
                    v = str(v)
                fields[f.name] = v if isinstance(v, (int, float, str, bool, list, dict, type(None))) else repr(v)
            d["fields"] = fields
        return d

    wanted = {
        "framewright.processors.pytorch_realesrgan": ["PyTorchESRGANConfig"],
        "framewright.processors.enhancement.super_resolution": ["SRBackend", "SRResult"],
        "framewright.processors.enhancement.denoising": ["DenoiserBackend", "DenoiseResult"],
        "framewright.processors.tap_denoise": ["TAPDenoiseConfig", "TAPDenoiseResult", "TAPDenoiser", "AutoTAPDenoiser",
                                               "MotionAdaptiveConfig", "MotionAdaptiveTAPDenoiser"],
        "framewright.plugins.base": ["ProcessorPlugin", "PluginMetadata"],
        "framewright.infrastructure.gpu.backends.base": ["Backend", "BackendCapabilities"],
        "framewright.processors.temporal_denoise": ["FlowField"],
        "framewright.processors.interpolation": ["InterpolationConfig", "FrameInterpolator"],
        "framewright.infrastructure.gpu.distributor": ["GPUStats", "DistributionPlan", "ProcessingResult", "GPUDistributor",
                                                       "MultiGPUProcessor"],
    }
    out, failed = {}, {}
    for mod, names in wanted.items():
        try:
            m = load_reference(mod)
        except Exception as e:  # noqa: BLE001 - an ordinary import error of the snapshot (missing sibling package)
            failed[mod] = f"{type(e).__name__}: {e}"
            continue
        for n in names:
            out[f"{mod.rsplit('.', 1)[1]}.{n}"] = describe(getattr(m, n))
    # module-level tables and the validation messages of the Real-ESRGAN module (pytorch_realesrgan.py:36-61, 264-275)
    pr = load_reference("framewright.processors.pytorch_realesrgan")
    tables = {"NCNN_TO_PYTORCH_MODEL": dict(pr.NCNN_TO_PYTORCH_MODEL),
              "convert_ncnn_model_name": {n: pr.convert_ncnn_model_name(n) for n in
                                          list(pr.NCNN_TO_PYTORCH_MODEL) + ["unknown", "", "RealESRGAN_x4plus"]},
              "validate": {}}
    for kw in ({}, {"model_name": "nope"}, {"scale_factor": 3}, {"scale_factor": 2}, {"model_name": "RealESRGAN_x2plus", "scale_factor": 2},
               {"model_name": "realesr-animevideov3"}, {"model_name": "RealESRGAN_x4plus_anime_6B"}):
        try:
            pr.PyTorchESRGANConfig(**kw).validate()
            tables["validate"][json.dumps(kw, sort_keys=True)] = None
        except Exception as e:  # noqa: BLE001
            tables["validate"][json.dumps(kw, sort_keys=True)] = f"{type(e).__name__}: {e}"
    (ROOT / "tests" / "golden" / "interfaces.json").write_text(json.dumps({"classes": out, "not_importable": failed, "tables": tables},
                                                                           indent=1, sort_keys=True))
    print("wrote tests/golden/interfaces.json;", len(out), "classes;", "not importable:", failed)


def aesrgan_attention() -> None:
    """Fixture set 7 - the reference's own ``AESRGAN`` (aesrgan_face.py:205-269) WITH live attention blocks (gamma != 0; its
    constructor's zero makes them the identity): scale 2 and 4, square and non-square inputs.  Weights are regenerated from
    seeds (synth.synthetic_rrdbnet_state / synthetic_attention_state), only inputs and outputs are stored.
    -> tests/golden/aesrgan_attention.npz"""
    from framewright_amd.synth import aesrgan_attention_positions, synthetic_attention_state
    a = load_reference("framewright.processors.aesrgan_face")
    rng = np.random.default_rng(909)
    out = {}
    for tag, (nb, na, scale, h, w, seed) in {"s4": (4, 2, 4, 12, 12, 31), "s2": (2, 2, 2, 9, 14, 32)}.items():
        sd = synthetic_rrdbnet_state(nb, 4, seed=seed)
        asd = synthetic_attention_state(nb, na, seed=seed + 100)
        net = a.AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=nb, scale=scale, num_attention=na).eval()
        pos = aesrgan_attention_positions(nb, na)
        ri = 0
        expect_attn = None
        for m in net.body:
            if isinstance(m, a.RRDB):
                _load_rrdb(m, sd, f"body.{ri}")
                expect_attn = ri if ri in pos else None
                ri += 1
            else:
                assert isinstance(m, a.AttentionBlock) and expect_attn is not None
                for name in ("query", "key", "value"):
                    conv = getattr(m, name)
                    conv.weight.data.copy_(torch.from_numpy(asd[f"attn.{expect_attn}.{name}.weight"]))
                    conv.bias.data.copy_(torch.from_numpy(asd[f"attn.{expect_attn}.{name}.bias"]))
                m.gamma.data.copy_(torch.from_numpy(asd[f"attn.{expect_attn}.gamma"]))
        assert ri == nb
        for name in ("conv_first", "conv_body", "conv_up1", "conv_hr", "conv_last") + (("conv_up2",) if scale >= 4 else ()):
            conv = getattr(net, name)
            conv.weight.data.copy_(torch.from_numpy(sd[name + ".weight"]))
            conv.bias.data.copy_(torch.from_numpy(sd[name + ".bias"]))
        x = rng.uniform(0, 1, size=(1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            y = net(torch.from_numpy(x).clone()).numpy()
        out[tag + "_cfg"] = np.array([nb, na, scale, seed], np.int64)
        out[tag + "_in"], out[tag + "_out"] = x, y
        print(tag, x.shape, "->", y.shape, "mean", float(y.mean()), "std", float(y.std()))
    dst = ROOT / "tests" / "golden" / "aesrgan_attention.npz"
    np.savez_compressed(dst, **out)
    print(f"wrote {dst} ({dst.stat().st_size / 1024:.0f} KiB)")


def interpolator_logic() -> None:
    """Fixture set 8 - processors/interpolation.py host logic evaluated by the reference's own ``FrameInterpolator`` (binary
    lookup patched, nothing downloaded) and by Pillow (the third-party package its motion-blur reduction calls, present here):
    pass counts (:488-499), InterpolationConfig validation messages (:66-75), model info tables (:501-528),
    ``detect_all_scene_changes`` on a PNG directory (:368-401; skimage is absent, so the reference itself takes its
    histogram branch), ``apply_motion_blur_reduction`` (:403-455) at three strengths.
    -> tests/golden/interpolator_reference.npz + .json"""
    import json
    import shutil
    import tempfile
    from unittest import mock

    from PIL import Image

    ip = load_reference("framewright.processors.interpolation")
    with mock.patch.object(shutil, "which", return_value="/usr/bin/true"):
        mk = lambda **kw: ip.FrameInterpolator(config=ip.InterpolationConfig(**kw))
        passes = {lvl: mk(smoothness=lvl)._get_pass_count() for lvl in ("low", "medium", "high")}
        fi = mk(scene_threshold=0.3)
        errors = {}
        for kw in ({"scene_threshold": 1.5}, {"scene_threshold": -0.1}, {"target_fps": 0}, {"smoothness": "ultra"}):
            try:
                ip.InterpolationConfig(**kw)
                errors[json.dumps(kw, sort_keys=True)] = None
            except Exception as e:  # noqa: BLE001
                errors[json.dumps(kw, sort_keys=True)] = f"{type(e).__name__}: {e}"
        info = {m: ip.FrameInterpolator.get_model_info(m) for m in ip.FrameInterpolator.SUPPORTED_MODELS + ["nope"]}
        models = ip.FrameInterpolator.list_available_models()
        # a 7-frame clip with two cuts: smooth drift, a hard cut to another scene, drift, a cut back
        rng = np.random.default_rng(11)
        yy, xx = np.mgrid[0:40, 0:56]
        scene_a = np.stack([120 + 80 * np.sin(xx / 9.0 + c) + 30 * np.cos(yy / 7.0) for c in range(3)], 2)
        scene_b = np.stack([60 + 150 * ((xx // 8 + yy // 8 + c) % 2) for c in range(3)], 2)
        clip = []
        for k, base in enumerate([scene_a, scene_a, scene_a, scene_b, scene_b, scene_a, scene_a]):
            clip.append(np.clip(np.roll(base, k, axis=1) + rng.normal(0, 3, base.shape), 0, 255).astype(np.uint8))
        with tempfile.TemporaryDirectory() as td:
            for k, f in enumerate(clip):
                Image.fromarray(f).save(Path(td) / f"frame_{k:08d}.png")
            bounds = fi.detect_all_scene_changes(Path(td))
            pair_flags = [bool(fi.detect_scene_change(Path(td) / f"frame_{k:08d}.png", Path(td) / f"frame_{k + 1:08d}.png"))
                          for k in range(len(clip) - 1)]
        img = np.clip(np.stack([128 + 90 * np.sin(xx / 3.0 + c) * np.cos(yy / 4.0) for c in range(3)], 2)
                      + rng.normal(0, 6, (40, 56, 3)), 0, 255).astype(np.uint8)
        sharp = {f"sharp_{s}": fi.apply_motion_blur_reduction(img, strength=s) for s in (0.5, 1.0, 2.0)}
    np.savez_compressed(ROOT / "tests" / "golden" / "interpolator_reference.npz", clip=np.stack(clip), img=img,
                        **{k.replace(".", "p"): v for k, v in sharp.items()})
    (ROOT / "tests" / "golden" / "interpolator_reference.json").write_text(json.dumps(
        {"pass_count": passes, "config_errors": errors, "model_info": info, "models": models, "scene_boundaries": bounds,
         "pair_flags": pair_flags, "has_skimage": bool(ip.HAS_SKIMAGE), "supported_models": ip.FrameInterpolator.SUPPORTED_MODELS,
         "supported_target_fps": ip.FrameInterpolator.SUPPORTED_TARGET_FPS}, indent=1, sort_keys=True))
    print("wrote tests/golden/interpolator_reference.{npz,json}; scene boundaries", bounds, "skimage", ip.HAS_SKIMAGE)


def gpu_distributor_logic() -> None:
    """Fixture set 9 - the five planners of infrastructure/gpu/distributor.py (:287-490) and ``get_optimal_distribution``
    (:472-507) evaluated by the reference's own ``GPUDistributor`` on hand-made device tables (its detector is bypassed by
    filling ``_devices`` / ``_stats``, the attributes its planners read).  -> tests/golden/gpu_distributor_reference.json"""
    import json
    d = load_reference("framewright.infrastructure.gpu.distributor")
    det = load_reference("framewright.infrastructure.gpu.detector")
    tables = {
        "one": [(0, 294912, 280000)],
        "eight_equal": [(i, 294912, 290000) for i in range(8)],
        "mixed": [(0, 24576, 20000), (1, 8192, 1000), (2, 16384, 16000)],
        "one_full": [(0, 294912, 0), (1, 294912, 100000)],
        "all_full": [(0, 8192, 0), (1, 4096, 0)],
    }
    timings = {"none": {}, "measured": {0: (12, 0.071), 1: (15, 0.140), 2: (11, 0.05)}, "few": {0: (3, 0.2)}}
    cases = []
    for tname, table in tables.items():
        for sname, tm in timings.items():
            for n in (0, 1, 7, 100, 301):
                gd = d.GPUDistributor()
                gd._devices = [det.DeviceInfo(index=i, name=f"gpu{i}", vendor=det.GPUVendor.AMD, total_memory_mb=t, free_memory_mb=f)
                               for i, t, f in table]
                for dev in gd._devices:
                    st = d.GPUStats(device_id=dev.index, vendor=dev.vendor, name=dev.name, total_memory_mb=dev.total_memory_mb)
                    if dev.index in tm:
                        cnt, avg = tm[dev.index]
                        st.frames_processed, st.avg_time_per_frame, st.total_time_seconds = cnt, avg, cnt * avg
                    gd._stats[dev.index] = st
                row = {"table": tname, "devices": table, "timing": sname, "stats": {str(k): list(v) for k, v in tm.items()}, "n": n, "plans": {}}
                for strat in d.DistributionStrategy:
                    try:
                        plan = gd.distribute_frames(n, strat)
                        row["plans"][strat.value] = {str(k): v for k, v in plan.gpu_workloads.items()}
                    except Exception as e:  # noqa: BLE001 - e.g. ZeroDivisionError of the load-balanced planner when no device has free memory
                        row["plans"][strat.value] = {"error": type(e).__name__}
                try:
                    opt = gd.get_optimal_distribution(n)
                    row["optimal"] = {str(k): v for k, v in opt.gpu_workloads.items()}
                except Exception as e:  # noqa: BLE001
                    row["optimal"] = {"error": type(e).__name__}
                cases.append(row)
    # unhealthy devices are skipped by distribute_frames (:262-270)
    gd = d.GPUDistributor(d.DistributionStrategy.ROUND_ROBIN)
    gd._devices = [det.DeviceInfo(index=i, name=f"gpu{i}", vendor=det.GPUVendor.AMD, total_memory_mb=1000, free_memory_mb=900) for i in range(3)]
    for dev in gd._devices:
        gd._stats[dev.index] = d.GPUStats(device_id=dev.index, vendor=dev.vendor, name=dev.name)
    gd.mark_device_unhealthy(1)
    unhealthy = {str(k): v for k, v in gd.distribute_frames(6).gpu_workloads.items()}
    (ROOT / "tests" / "golden" / "gpu_distributor_reference.json").write_text(json.dumps({"cases": cases, "unhealthy_rr": unhealthy}))
    print("wrote tests/golden/gpu_distributor_reference.json;", len(cases), "cases")


def assign_frames_logic() -> None:
    """Fixture set 6 - `MultiGPUDistributor._assign_frames` (utils/multi_gpu.py:780-870) evaluated by the reference itself for
    every LoadBalanceStrategy on synthetic GPUInfo lists.  -> tests/golden/assign_frames.json"""
    import json
    mg = load_reference("framewright.utils.multi_gpu")
    rng = np.random.default_rng(77)
    cases = []
    for n_gpu in (1, 2, 3, 8):
        for n_frames in (0, 1, 7, 40, 301):
            for variant in range(3):
                gpus = []
                for i in range(n_gpu):
                    total = int(rng.choice([24576, 81920, 294912]))
                    free = 0 if variant == 2 else int(rng.integers(0, total + 1))
                    util = 100.0 if variant == 2 else float(rng.integers(0, 101))
                    gpus.append({"id": i if variant != 1 else (n_gpu - 1 - i) * 2, "name": f"g{i}", "total_vram_mb": total,
                                 "free_vram_mb": free, "utilization_pct": util})
                for strat in mg.LoadBalanceStrategy:
                    d = mg.MultiGPUDistributor.__new__(mg.MultiGPUDistributor)
                    d.strategy = strat
                    plan = d._assign_frames(list(range(n_frames)), [mg.GPUInfo(**g) for g in gpus])
                    cases.append({"gpus": gpus, "n_frames": n_frames, "strategy": strat.value,
                                  "plan": {str(k): v for k, v in plan.items()}})
    (ROOT / "tests" / "golden" / "assign_frames.json").write_text(json.dumps(cases))
    print("wrote tests/golden/assign_frames.json;", len(cases), "cases")


def face_restorer_logic() -> None:
    """Fixture set 9 - the host side of AESRGANFaceRestorer (reference src/framewright/processors/aesrgan_face.py:51-136, 270-760):
    the interfaces (fields, defaults, method parameters, enum values, validation messages, class constants) read off the reference's
    own classes, and `_extract_face` RUN on the reference's own instance (pure index arithmetic: no cv2 needed).  `_paste_face_back`
    and the detectors call cv2, which this image does not have: they cannot be run, their restatement (oracle/face_ref.py) stays
    unpinned.  -> tests/golden/face_reference.json"""
    import dataclasses
    import enum
    import inspect
    import json
    import tempfile
    a = load_reference("framewright.processors.aesrgan_face")

    def describe(cls):
        d = {"methods": {}}
        for name, fn in inspect.getmembers(cls, predicate=inspect.isfunction):
            if name.startswith("__") and name != "__init__":
                continue
            d["methods"][name] = list(inspect.signature(fn).parameters)
        for name, prop in inspect.getmembers(cls, lambda o: isinstance(o, property)):
            d["methods"][name] = ["<property>"]
        if dataclasses.is_dataclass(cls):
            fields = {}
            for f in dataclasses.fields(cls):
                v = f.default if f.default is not dataclasses.MISSING else "<required>"
                if isinstance(v, enum.Enum):
                    v = v.value
                fields[f.name] = v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)
            d["fields"] = fields
        return d

    out = {"classes": {n: describe(getattr(a, n)) for n in ("AESRGANFaceConfig", "FaceBox", "AESRGANFaceResult", "AESRGANFaceRestorer", "FaceDetector")},
           "FaceDetectorType": {m.name: m.value for m in a.FaceDetectorType},
           "constants": {"MODEL_FILE": a.AESRGANFaceRestorer.MODEL_FILE, "DEFAULT_MODEL_DIR_tail": list(a.AESRGANFaceRestorer.DEFAULT_MODEL_DIR.parts[-3:])},
           "factory_params": list(inspect.signature(a.create_aesrgan_restorer).parameters), "validate": {}, "extract": []}
    for kw in ({}, {"detection_threshold": 1.5}, {"enhancement_strength": -0.1}, {"upscale_factor": 3}, {"face_detector": "opencv"}, {"upscale_factor": 4}):
        try:
            c = a.AESRGANFaceConfig(**kw)
            out["validate"][json.dumps(kw, sort_keys=True)] = {"ok": True, "face_detector": c.face_detector.value}
        except Exception as e:  # noqa: BLE001
            out["validate"][json.dumps(kw, sort_keys=True)] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    with tempfile.TemporaryDirectory() as td:
        r = a.AESRGANFaceRestorer(a.AESRGANFaceConfig(), model_dir=Path(td))
        out["backend_without_opencv"] = r._backend          # the reference disables itself without cv2: None
        rng = np.random.default_rng(5)
        for (h, w, box, pad) in [(120, 160, (40, 30, 100, 90), 0.3), (120, 160, (0, 0, 50, 40), 0.3), (120, 160, (130, 90, 160, 120), 0.3),
                                 (64, 48, (10, 12, 31, 45), 0.5), (64, 48, (5, 5, 6, 6), 0.3), (200, 300, (100, 50, 223, 181), 0.0)]:
            frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            fb = a.FaceBox(x1=box[0], y1=box[1], x2=box[2], y2=box[3], confidence=0.9)
            crop, region = r._extract_face(frame, fb, padding=pad)
            out["extract"].append({"h": h, "w": w, "box": list(box), "padding": pad, "region": [int(v) for v in region], "crop_shape": list(crop.shape),
                                   "crop_sum": int(crop.astype(np.int64).sum()), "width": fb.width, "height": fb.height, "center": list(fb.center),
                                   "frame_seed_index": len(out["extract"])})
    (ROOT / "tests" / "golden" / "face_reference.json").write_text(json.dumps(out, indent=1, sort_keys=True))
    print("wrote tests/golden/face_reference.json;", len(out["extract"]), "extract cases; backend without cv2:", out["backend_without_opencv"])


if __name__ == "__main__":
    main()
    host_logic()
    tap_logic()
    tile_and_flow_logic()
    interfaces()
    assign_frames_logic()
    aesrgan_attention()
    interpolator_logic()
    gpu_distributor_logic()
    face_restorer_logic()
