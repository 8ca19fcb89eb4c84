#!/usr/bin/env python3
"""Throughput of the default path with one frame at a time against several frames in flight
(one mrt_ctx and one stream per frame in flight, frames handed out in rotation).  GPU box only.

usage: time_overlap.py [teatime|soup] XRES YRES [STEPS [FLAGS]]     (FLAGS: MRT_FLAG_* as an integer)
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
kind, xres, yres = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
flags = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
text = mkscene.teatime(xres, yres) if kind == "teatime" else mkscene.soup(xres, yres)
scene = mrt.LoadedScene(text=text).copy()
dev = torch.device("cuda", 0)


def run(n_flight):
    ctxs, streams, outs = [], [], []
    for _ in range(n_flight):
        c = mrt.Context(device=0, flags=flags)
        c.set_scene(scene)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(dev))
        outs.append(torch.empty(xres * yres, dtype=torch.int32, device=dev))

    def frames(n):
        for k in range(n):
            b = k % n_flight
            ctxs[b].render_bands_device(64, 0, 1, outs[b].data_ptr(), outs[b].numel() * 4,
                                        stream=streams[b].cuda_stream)

    frames(10)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        frames(steps)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    ref = outs[0].clone()
    same = all(bool((o == ref).all()) for o in outs)
    for c in ctxs:
        c.close()
    return best, same


for n in (1, 2, 3, 1, 2):
    ms, same = run(n)
    print(f"frames_in_flight={n} ms_per_frame={ms:.4f} mrays_per_s={xres * yres / ms * 1e-3:.1f} same_pixels={same}", flush=True)
