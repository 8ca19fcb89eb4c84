#!/usr/bin/env python3
"""Frames per second of one mrt_ctx (a frame at a time) against an mrt_sequence (frames in flight)
on the teatime mesh at several resolutions, and that both write the same bytes.  GPU box only.

usage: time_sequence.py [XRESxYRES ...]
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
import mrt_pkg  # noqa: E402
import mkscene  # noqa: E402

mrt = mrt_pkg.load()
dev = torch.device("cuda", 0)
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or \
    [(400, 400), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)]
for xres, yres in sizes:
    scene = mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy()
    n = xres * yres * 4
    steps = max(10, min(300, int(120e6 / (xres * yres))))
    bufs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    ctx = mrt.Context(device=0)
    ctx.set_scene(scene)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def one(k):
        ctx.render_bands_device(64, 0, 1, bufs[k % 4].data_ptr(), n, stream=stream)

    res = {}
    for label, in_flight in (("ctx", 0), ("seq2", 2), ("seq3", 3)):
        seq = None
        if in_flight:
            seq = mrt.Sequence(device=0, in_flight=in_flight)
            seq.set_scene(scene)
            frame = lambda k: seq.submit(64, 0, 1, bufs[k % 4].data_ptr(), n)      # noqa: E731
        else:
            frame = one
        for k in range(8):
            frame(k)
        torch.cuda.synchronize(dev)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(steps):
                frame(k)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / steps * 1e3
            best = dt if best is None else min(best, dt)
        res[label] = (best, bufs[0].clone())
        if seq is not None:
            seq.close()
    ctx.close()
    same = all(bool((res[k][1] == res["ctx"][1]).all()) for k in res)
    print(f"{xres}x{yres}: " + "  ".join(f"{k} {v[0]:.4f} ms ({xres * yres / v[0] * 1e-3:.0f} Mrays/s)" for k, v in res.items()) +
          f"  same_bytes={same}", flush=True)
