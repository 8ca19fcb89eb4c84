#!/usr/bin/env python3
"""One-off soak (GPU box) of mrt_stage_intersect: random objects in mixed order (with duplicates
further down the array), random rays with an origin and a direction of their own, scene spans from
1e-3 to 1e4 and random (t_lower, t_upper) pairs, every 80-byte record against the oracle's per-ray
loop (kernel.cl:305-379).  usage: soak_stages.py [N_CASES] [SEED]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools", "tests", "oracle"):
    sys.path.insert(0, str(ROOT / d))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import mrt_pkg  # noqa: E402
import oracle_py  # noqa: E402
import test_gpu_stages as ts  # noqa: E402

mrt = mrt_pkg.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = mrt.Context(device=0)
FLT_MAX = float(np.finfo(np.float32).max)
bad = 0
pairs = 0
switched = 0          # calls in which at least one wave took up the filter with the fourth determinant
t0 = time.time()
for it in range(n_cases):
    span = float(10.0 ** rng.uniform(-3, 4))
    n_tri = int(rng.integers(0, 1200))
    n_sph = int(rng.integers(0 if n_tri else 1, 200))
    n_rays = int(rng.integers(1, 12000))
    objs = ts.random_objects(mrt, rng, n_tri, n_sph, span=span, tri_size=span * float(10.0 ** rng.uniform(-2, 0)))
    sph = np.flatnonzero(objs["is_circle"])
    objs["radius"][sph] *= span / 3.0
    if len(objs) > 8:
        objs[-1] = objs[2]
        objs[-3] = objs[5]
    rays = ts.random_rays(mrt, rng, n_rays, span=span)
    if it % 3 == 0:                                   # rays from one point, as a camera's
        rays["origin"][:, :3] = rng.uniform(-span, span, 3)
    if it % 5 == 0:                                   # ... or of one direction
        rays["direction"][:, :3] = rays["direction"][0, :3]
    for _ in range(2):
        lo = float(rng.choice([0.0, 0.0, -1.0 * span, 0.25 * span, -np.inf]))
        hi = float(rng.choice([FLT_MAX, FLT_MAX, np.inf, 2.0 * span, 0.5 * span]))
        g = ctx.stage_intersect(objs, rays, hi, lo)
        switched += 1 if ctx.stage_t_switches() else 0
        o, _ = oracle_py.stage_intersect(objs, rays, hi, lo)
        pairs += len(objs) * n_rays
        try:
            ts.same_records(g, o)
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH case {it}: span {span:.3g} tri {n_tri} sph {n_sph} rays {n_rays} t in ({lo}, {hi}): {str(e)[:200]}", flush=True)
    if it % 20 == 0:
        print(f"case {it}: span {span:.3g}, {n_tri} triangles, {n_sph} spheres, {n_rays} rays; {pairs:.3g} pairs so far, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print(f"cases: {n_cases} pairs: {pairs} calls with waves that switched filters: {switched} mismatches: {bad}")
sys.exit(1 if bad else 0)
