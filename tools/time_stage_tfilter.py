#!/usr/bin/env python3
"""mrt_stage_intersect with and without the fourth determinant (Nt, the sign of t) in the
conservative filter, on rays for which it matters least and most: teatime's primary rays (every
triangle lies in front of the camera) and rays that start inside the mesh's bounding box with
random directions (half of what their lines meet lies behind them).  Run once per setting of
MRT_STAGE_T_FILTER (0 / 1: read once per process).  GPU box only."""
import os
import sys
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
sc = scene_py.load_scene(ROOT / "scenes" / "teatime.json")
pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 400)
ctx = mrt.Context(device=0)
FLT_MAX = float(np.finfo(np.float32).max)
primary = ctx.stage_viewrays(cam, ctx.stage_uv(pos, vp), sc.struct.focal)
tri = objs[objs["is_triangle"] != 0]
verts = np.concatenate([tri[k][:, :3] for k in ("a", "b", "c")])
lo, hi = verts.min(axis=0), verts.max(axis=0)
rng = np.random.default_rng(5)
inside = np.zeros(len(primary), mrt.CL_DTYPES["ray"])
inside["origin"][:, :3] = rng.uniform(lo, hi, (len(primary), 3)).astype(np.float32)
d = rng.normal(size=(len(primary), 3))
inside["direction"][:, :3] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
# ... and where it matters most: a stack of 4032 sheets, each of which every ray's line meets, all of them behind the rays
from test_gpu_stages import stacked_sheets  # noqa: E402
sheets = stacked_sheets(mrt, 4032)
away = np.zeros(len(primary), mrt.CL_DTYPES["ray"])
away["origin"][:, 0:2] = rng.uniform(-2, 2, (len(primary), 2))
away["origin"][:, 2] = 5.0
away["direction"][:, 0:2] = rng.uniform(-0.01, 0.01, (len(primary), 2))
away["direction"][:, 2] = 1.0
for name, rays in (("primary", primary), ("inside", inside), ("away_from_sheets", away)):
    if name == "away_from_sheets":
        objs = sheets
    ms = []
    for k in range(22):
        xs = ctx.stage_intersect(objs, rays, FLT_MAX, 0.0)
        if k >= 2:
            ms.append(ctx.stage_kernel_ms())
    ms.sort()
    print(f"T_FILTER={os.environ.get('MRT_STAGE_T_FILTER', '1')} {name}: kernels best {ms[0]:.3f} median {ms[len(ms) // 2]:.3f} ms, "
          f"waves that took up the fourth determinant {ctx.stage_t_switches()}, hits {(xs['intersects'] != 0).mean():.3f}, checksum {int(xs['t_'].view(np.uint32).astype(np.uint64).sum())}", flush=True)
ctx.close()
