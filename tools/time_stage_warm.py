#!/usr/bin/env python3
"""mrt_stage_intersect's kernels after an idle gap and right behind other work.

The bench line times the frame kernels on a busy GPU (every timed frame is enqueued behind an
untimed one) but the stage call the way a host program makes it: one blocking call after another,
the GPU idle while the host packs, copies and scatters.  The GPU's clock follows its load, so a
0.3 ms kernel after such a gap need not run at the clock the roofline's peak assumes.  This
script times the same call (teatime's 160 000 primary rays x 4034 objects) three ways: as the
bench does; right after 25 ms of frame kernels of another context (two MRT_FLAG_FULL_TEST frames
at 3840x2160, waited for, so that nothing runs beside the stage call's kernels); and again
after a pause of 50 ms.  GPU box only."""
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
import scene_py  # noqa: E402
from test_gpu_stages import make_inputs  # noqa: E402

mrt = mrt_pkg.load()
sc = scene_py.load_scene(ROOT / "scenes" / "teatime.json")
pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 400)
ctx = mrt.Context(device=0)
FLT_MAX = float(np.finfo(np.float32).max)
rays = ctx.stage_viewrays(cam, ctx.stage_uv(pos, vp), sc.struct.focal)
xs = ctx.stage_intersect(objs, rays, FLT_MAX, 0.0)

big = mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy()
warm = mrt.Context(device=0, flags=mrt.FLAG_FULL_TEST)
warm.set_scene(big)
frame = torch.empty(3840 * 2160 * 4, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream


def busy(frames):
    for _ in range(frames):
        warm.render_bands_device(64, 0, 1, frame.data_ptr(), frame.numel(), stream=stream)
    torch.cuda.synchronize()


def series(name, before):
    ms = []
    for k in range(14):
        before()
        ctx.stage_intersect(objs, rays, FLT_MAX, 0.0, out=xs)
        if k >= 2:
            ms.append(ctx.stage_kernel_ms())
    ms.sort()
    pairs = len(rays) * len(objs)
    print(f"{name}: kernels best {ms[0]:.3f} median {ms[len(ms) // 2]:.3f} worst {ms[-1]:.3f} ms = "
          f"{pairs / ms[len(ms) // 2] * 1e-9:.3f}e12 tests/s at the median", flush=True)


for rnd in range(2):
    series("call after call (the bench's way)", lambda: None)
    series("right behind two full-test 4K frames", lambda: busy(2))
    series("after a pause of 50 ms", lambda: time.sleep(0.05))
ctx.close()
warm.close()
