#!/usr/bin/env python3
"""One-off soak (GPU box): random scenes of the fuzz families at larger sizes, every kernel
against the exact-only path.  usage: soak_kernels.py [N_SCENES] [SEED]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools", "tests", "oracle"):
    sys.path.insert(0, str(ROOT / d))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import mrt_pkg  # noqa: E402
import test_gpu_prefilter_fuzz as fz  # noqa: E402

mrt = mrt_pkg.load()
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kinds = ["generic", "needle", "edge_on", "subpixel", "huge", "behind", "far", "tiny", "micro", "denormal", "giant"]
kinds_of_renderer = {"default": 0, "wide": mrt.FLAG_TRACE_WIDE, "full": mrt.FLAG_FULL_TEST, "cull": "cull", "block": "block", "grid": "grid"}
ctxs = {k: mrt.renderer(device=0, kind=v) for k, v in kinds_of_renderer.items()}
exact = mrt.Context(device=0, flags=mrt.FLAG_NO_PREFILTER)
bad = 0
t0 = time.time()
for it in range(n_scenes):
    kind = kinds[it % len(kinds)]
    proj = mrt.PROJ_PERSP if (it // len(kinds)) % 2 == 0 else mrt.PROJ_ORTHO
    sc = fz._scene(mrt, rng, kind, proj)
    # larger frame than the unit test uses: several wave tiles in both directions
    sc = sc.with_resolution(int(rng.integers(200, 900)), int(rng.integers(150, 600)))
    exact.set_scene(sc)
    _, ids_e, t_e, _ = exact.render(want_ids=True)
    for name, ctx in ctxs.items():
        ctx.set_scene(sc)
        _, ids, t, _ = ctx.render(want_ids=True)
        if not (np.array_equal(ids, ids_e) and np.array_equal(t.view(np.uint32), t_e.view(np.uint32))):
            bad += 1
            print(f"MISMATCH scene {it} kind {kind} proj {proj} kernel {name}: {(ids != ids_e).sum()} ids", flush=True)
    if it % 20 == 0:
        print(f"scene {it} ({kind}) hits {(ids_e >= 0).mean():.3f} elapsed {time.time() - t0:.0f}s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
