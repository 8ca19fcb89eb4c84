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
    kms = ctx.stage_kernel_ms()
    col = ctx.stage_phong(xs, phong, bg, cam, lights); t.append(time.perf_counter())
    ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print(f"stages {name} 400x400: uv {ms[0]:.2f} rays {ms[1]:.2f} intersect {ms[2]:.2f} phong {ms[3]:.2f} total {sum(ms):.2f} ms; intersect kernels {kms:.3f} ms", flush=True)
# ... and with the caller's result arrays kept from call to call, as a host program holds them (a fresh
# numpy array per call, above, pays a page fault per 4 KB the library first writes to)
for rep in range(5):
    t = [time.perf_counter()]
    rays = ctx.stage_viewrays(cam, uv, focal, out=rays); t.append(time.perf_counter())
    xs = ctx.stage_intersect(objs, rays, FLT_MAX, 0.0, out=xs); t.append(time.perf_counter())
    col = ctx.stage_phong(xs, phong, bg, cam, lights, out=col); t.append(time.perf_counter())
    ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print(f"stages {name} 400x400, arrays reused: rays {ms[0]:.2f} intersect {ms[1]:.2f} phong {ms[2]:.2f} ms", flush=True)
# the device side of stage_intersect alone, warm: every (ray, object) pair in kms milliseconds
import os  # noqa: E402
t_lower = float(os.environ.get("MRT_TIME_STAGES_T_LOWER", "0"))
kk = []
for rep in range(30):
    ctx.stage_intersect(objs, rays, FLT_MAX, t_lower)
    kk.append(ctx.stage_kernel_ms())
kk.sort()
pairs = len(rays) * len(objs)
print(f"stage_intersect kernels {name}: {len(rays)} rays x {len(objs)} objects: best {kk[0]:.3f} ms, median {kk[len(kk) // 2]:.3f} ms "
      f"= {pairs / kk[len(kk) // 2] * 1e-9:.3f}e12 tests/s", flush=True)
ctx.set_scene(sc)
for rep in range(3):
    t0 = time.perf_counter()
    img, stats = ctx.render()
    print(f"fused  {name} 400x400: {(time.perf_counter() - t0) * 1e3:.2f} ms (kernel {stats.kernel_ms:.3f})", flush=True)
ctx.close()
