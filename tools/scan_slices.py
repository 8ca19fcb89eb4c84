#!/usr/bin/env python3
"""trace_kernel time against the number of triangle-list slices (the override MRT_FORCE_SLICES also sets) for both lane
shapes.  GPU box only.  usage: scan_slices.py [teatime|soup] XRES YRES [--world W] [--shape narrow|wide] [slices ...]
(--world W: rank 0 of W alone, balanced bands)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402,F401
import mrt_pkg  # noqa: E402
import mkscene  # noqa: E402

mrt = mrt_pkg.load()
kind, xres, yres = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
world, shapes = 1, ("narrow", "wide")
while rest and rest[0].startswith("--"):
    if rest[0] == "--world":
        world = int(rest[1])
    elif rest[0] == "--shape":
        shapes = (rest[1],)
    rest = rest[2:]
counts = [int(a) for a in rest] or [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 128]
band = mrt.balanced_band_rows(yres, world) if world > 1 else 64
text = mkscene.teatime(xres, yres) if kind == "teatime" else mkscene.soup(xres, yres)
scene = mrt.LoadedScene(text=text).copy()
out = torch.empty(xres * mrt.band_rows(yres, band, 0, world), dtype=torch.int32, device="cuda:0")
for name, flag in (("narrow", mrt.FLAG_TRACE_NARROW), ("wide", mrt.FLAG_TRACE_WIDE)):
    if name not in shapes:
        continue
    ctx = mrt.Context(device=0, flags=flag)
    ctx.set_scene(scene)
    for n in [0] + counts:
        ctx.force_slices(n)
        ms, slices = [], 0
        for _ in range(30):
            st = ctx.render_bands_device(band, 0, world, out.data_ptr(), out.numel() * 4, timed=True)
            ms.append(st.trace_ms + st.shade_ms)
            slices = st.prim_slices
        ms.sort()
        print(f"{kind} {xres}x{yres} w{world} {name} asked={n or 'auto'} slices={slices} best={ms[0]:.3f} median={ms[len(ms) // 2]:.3f}", flush=True)
    ctx.close()
