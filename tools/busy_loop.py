#!/usr/bin/env python3
"""Keeps the GPU busy with one kind of work for a number of seconds, for tools/sample_clocks.sh to
read the clocks and the power beside it.  usage: busy_loop.py full|default|stage SECONDS [MARKER_FILE]
  full:    MRT_FLAG_FULL_TEST frames of teatime at 3840x2160, enqueued back to back
  default: default-path frames of the same scene
  stage:   mrt_stage_intersect calls (160 000 rays x 4034 objects), one after the other
GPU box only."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools", "tests", "oracle"):
    sys.path.insert(0, str(ROOT / d))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mkscene  # noqa: E402
import mrt_pkg  # noqa: E402

mrt = mrt_pkg.load()
kind, seconds = sys.argv[1], float(sys.argv[2])
marker = Path(sys.argv[3]) if len(sys.argv) > 3 else None      # touched when the loop is about to start
n = 0


def begin():
    if marker:
        marker.write_text("running")
    return time.time() + seconds


if kind in ("full", "default"):
    ctx = mrt.Context(device=0, flags=mrt.FLAG_FULL_TEST if kind == "full" else 0)
    ctx.set_scene(mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy())
    frame = torch.empty(3840 * 2160 * 4, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    per = 4 if kind == "full" else 40
    t_end = begin()
    while time.time() < t_end:
        for _ in range(per):
            ctx.render_bands_device(64, 0, 1, frame.data_ptr(), frame.numel(), stream=stream)
        torch.cuda.synchronize()
        n += per
else:
    import scene_py
    from test_gpu_stages import make_inputs
    sc = scene_py.load_scene(ROOT / "scenes" / "teatime.json")
    pos, vp, cam, objs, _, _ = make_inputs(mrt, sc, 400)
    ctx = mrt.Context(device=0)
    rays = ctx.stage_viewrays(cam, ctx.stage_uv(pos, vp), sc.struct.focal)
    xs = None
    t_end = begin()
    while time.time() < t_end:
        xs = ctx.stage_intersect(objs, rays, float(np.finfo(np.float32).max), 0.0, out=xs)
        n += 1
ctx.close()
print(f"busy_loop {kind}: {n} frames / calls in {seconds:.0f} s", flush=True)
