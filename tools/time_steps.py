#!/usr/bin/env python3
"""Where the wall time of a short timed region goes: K frames enqueued behind a synchronize, with
an event after every frame.  Prints the host time of each enqueue, the GPU time of each frame and
the wall time of the whole region.  GPU box only.

usage: time_steps.py [XRES YRES [STEPS [WARMUP]]]
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
xres = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
yres = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 5
scene = mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy()
dev = torch.device("cuda", 0)
ctx = mrt.Context(device=0)
ctx.set_scene(scene)
out = torch.empty(xres * yres, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)


def frame():
    ctx.render_bands_device(64, 0, 1, out.data_ptr(), out.numel() * 4, stream=stream.cuda_stream)


for rep in range(3):
    for _ in range(warmup):
        frame()
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(steps):
        frame()
        ev[k + 1].record(stream)
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) * 1e3
    gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
    print(f"rep {rep}: wall {wall:.3f} ms = {wall / steps:.4f} per step; gpu first-to-last event "
          f"{ev[0].elapsed_time(ev[steps]):.3f} ms", flush=True)
    print("  gpu ms per frame:", " ".join(f"{g:.3f}" for g in gpu), flush=True)
    print("  host enqueue done at ms:", " ".join(f"{h:.3f}" for h in host), flush=True)
ctx.close()
