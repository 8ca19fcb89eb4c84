#!/usr/bin/env python3
"""Wall time of the per-stage entry points (the call shape of the reference's cl_* helpers) on
one shipped scene at its native 400x400, next to the fused path.  GPU box only."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools", "tests", "oracle"):
    sys.path.insert(0, str(ROOT / d))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import mrt_pkg  # noqa: E402
import scene_py  # noqa: E402
from test_gpu_stages import make_inputs  # noqa: E402

mrt = mrt_pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "teatime"
sc = scene_py.load_scene(ROOT / "scenes" / f"{name}.json")
st = sc.struct
pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 400)
ctx = mrt.Context(device=0)
FLT_MAX = np.finfo(np.float32).max
focal = None if st.projection == mrt.PROJ_ORTHO else st.focal
bg = tuple(st.background) + (1.0,)
for rep in range(3):
    t = [time.perf_counter()]
    uv = ctx.stage_uv(pos, vp); t.append(time.perf_counter())
    rays = ctx.stage_viewrays(cam, uv, focal); t.append(time.perf_counter())
    xs = ctx.stage_intersect(objs, rays, FLT_MAX, 0.0); t.append(time.perf_counter())
    col = ctx.stage_phong(xs, phong, bg, cam, lights); t.append(time.perf_counter())
    ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print(f"stages {name} 400x400: uv {ms[0]:.2f} rays {ms[1]:.2f} intersect {ms[2]:.2f} phong {ms[3]:.2f} total {sum(ms):.2f} ms", flush=True)
ctx.set_scene(sc)
for rep in range(3):
    t0 = time.perf_counter()
    img, stats = ctx.render()
    print(f"fused  {name} 400x400: {(time.perf_counter() - t0) * 1e3:.2f} ms (kernel {stats.kernel_ms:.3f})", flush=True)
ctx.close()
