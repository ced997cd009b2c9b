#!/usr/bin/env python3
"""Is a path bound by the host's launch rate?  For IFNet (1080p pair), NAFNet (1080p frame) and Restormer (512x512 tile): the time
the host needs to SUBMIT n forwards on one stream (no synchronisation inside) next to the time until the GPU has finished them, and
the same with one submitting thread per stream (engine clones)."""
import sys
import threading
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd import restormer as RS
from framewright_amd import rife as RF
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state


NCLONE = int(__import__('os').environ.get('FW_SUBMIT_CLONES', '3'))


def measure(name, engines, call, n):
    """engines: clones; call(engine, i) submits forward i."""
    for e in engines:
        call(e, 0)
    torch.cuda.synchronize()
    # one stream, one thread
    t0 = time.perf_counter()
    for i in range(n):
        call(engines[0], i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: one stream: submit {1e3 * (t1 - t0) / n:.3f} ms, done {1e3 * (t2 - t0) / n:.3f} ms per forward", flush=True)
    # one thread per stream
    for k in (2, 3, 4, 6):
        if len(engines) < k:
            break
        streams = [torch.cuda.Stream() for _ in range(k)]
        threads = not __import__('os').environ.get('FW_SUBMIT_NOTHREADS')
        def work(j):
            with torch.cuda.stream(streams[j]):
                for i in range(j, n, k):
                    call(engines[j], i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j,)) for j in range(k)] if threads else []
        for t in th:
            t.start()
        for t in th:
            t.join()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if threads:
            print(f"{name}: {k} threads / streams: submit {1e3 * (t1 - t0) / n:.3f} ms, done {1e3 * (t2 - t0) / n:.3f} ms per forward", flush=True)
        # same streams, ONE submitting thread (what the engines do today)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % k]):
                call(engines[i % k], i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: {k} streams, one thread: submit {1e3 * (t1 - t0) / n:.3f} ms, done {1e3 * (t2 - t0) / n:.3f} ms per forward", flush=True)


def main():
    which = sys.argv[1:] or ["ifnet", "nafnet", "restormer"]
    if "ifnet" in which:
        fr = synthetic_frames(2, 1080, 1920, seed=3)
        a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
        e0 = RF.IFNetEngine("f16"); e0.load_state_dict(synthetic_ifnet_state())
        engs = [e0] + [e0.clone() for _ in range(NCLONE - 1)]
        outs = [torch.empty_like(a) for _ in range(NCLONE)]
        measure("ifnet 1080p pair", engs, lambda e, i: e.interpolate_device(a, b, out=outs[engs.index(e)]), 60)
    if "nafnet" in which:
        f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda()
        e0 = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); e0.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
        engs = [e0] + [e0.clone() for _ in range(NCLONE - 1)]
        outs = [torch.empty_like(f) for _ in range(NCLONE)]
        measure("nafnet 1080p", engs, lambda e, i: e.denoise_device(f, out=outs[engs.index(e)]), 24)
    if "restormer" in which:
        f = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda()
        e0 = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); e0.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
        engs = [e0] + [e0.clone() for _ in range(NCLONE - 1)]
        outs = [torch.empty_like(f) for _ in range(NCLONE)]
        measure("restormer 512 tile", engs, lambda e, i: e.denoise_device(f, out=outs[engs.index(e)]), 18)


if __name__ == "__main__":
    main()
