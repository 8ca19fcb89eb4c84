#!/usr/bin/env python3
"""trace_kernel time against the number of triangle-list slices (the override MRT_FORCE_SLICES also sets) for both lane
shapes.  GPU box only.  usage: scan_slices.py [teatime|soup] XRES YRES [slices ...]"""
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
counts = [int(a) for a in sys.argv[4:]] or [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 128]
text = mkscene.teatime(xres, yres) if kind == "teatime" else mkscene.soup(xres, yres)
scene = mrt.LoadedScene(text=text).copy()
out = torch.empty(xres * yres, dtype=torch.int32, device="cuda:0")
for name, flag in (("narrow", mrt.FLAG_TRACE_NARROW), ("wide", mrt.FLAG_TRACE_WIDE)):
    ctx = mrt.Context(device=0, flags=flag)
    ctx.set_scene(scene)
    for n in [0] + counts:
        ctx.force_slices(n)
        ms, slices = [], 0
        for _ in range(30):
            st = ctx.render_bands_device(64, 0, 1, out.data_ptr(), out.numel() * 4, timed=True)
            ms.append(st.trace_ms + st.shade_ms)
            slices = st.prim_slices
        ms.sort()
        print(f"{kind} {xres}x{yres} {name} asked={n or 'auto'} slices={slices} best={ms[0]:.3f} median={ms[len(ms) // 2]:.3f}", flush=True)
    ctx.close()
