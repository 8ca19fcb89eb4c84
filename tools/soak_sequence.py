#!/usr/bin/env python3
"""One-off soak (GPU box) of mrt_sequence: random scenes, random lists of nearby views with
repeats and changes of resolution, one to four frames in flight, frames to device buffers and to
host memory mixed; every frame against a fresh context's render of its view.
usage: soak_sequence.py [N_SEQUENCES] [SEED]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools", "tests", "oracle"):
    sys.path.insert(0, str(ROOT / d))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mrt_pkg  # noqa: E402
import scene_py  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

mrt = mrt_pkg.load()
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fresh = mrt.Context(device=0)
bad = 0
frames = 0
t0 = time.time()
for it in range(n_seq):
    big = it % 7 == 0                      # now and then frames of 2 Mpixel and more: the slots' other lane shape
    w0, h0 = (int(rng.integers(1500, 2200)), int(rng.integers(1000, 1400))) if big else \
             (int(rng.integers(60, 700)), int(rng.integers(40, 500)))
    base = tp._random_scene(mrt, rng, mrt.PROJ_PERSP, w0, h0, int(rng.integers(1, 1500)), int(rng.integers(0, 60)))
    b = base.struct
    eye0, wv, vv = np.array(list(b.eye)), np.array(list(b.w)), np.array(list(b.v))

    def view(k):
        u, v, w = scene_py.camera_basis(list(-wv + rng.normal(scale=0.1, size=3)), list(vv + rng.normal(scale=0.1, size=3)))
        xres, yres = (w0, h0) if rng.random() < 0.7 else (max(1, w0 - int(rng.integers(0, 50))), max(1, h0 - int(rng.integers(0, 40))))
        return mrt.Scene(eye=eye0 + rng.normal(scale=0.3, size=3), u=u, v=v, w=w, xres=xres, yres=yres,
                         left=-1.2, right=0.9 + 0.02 * k, top=0.7, bottom=-1.1,
                         projection=mrt.PROJ_PERSP if rng.random() < 0.8 else mrt.PROJ_ORTHO, focal=float(b.focal),
                         shader=b.shader, ka=b.ka, kd=b.kd, ks=b.ks, ambient=list(b.ambient), background=list(b.background),
                         sph=base.sph, sph_mat=base.sph_mat, tri=base.tri, tri_mat=base.tri_mat, mat=base.mat, light=base.light)

    views = [None if (k == 0 or rng.random() < 0.25) else view(k) for k in range(int(rng.integers(4, 12)))]
    seq = mrt.Sequence(device=0, in_flight=int(rng.integers(1, 5)))
    seq.set_scene(base)
    outs, current = [], base
    for v in views:
        current = v if v is not None else current
        st = current.struct
        if rng.random() < 0.5:
            out = mrt.host_array((st.yres, st.xres, 4)) if rng.random() < 0.7 else np.zeros((st.yres, st.xres, 4), np.uint8)
            seq.submit_host(out, view=v)
        else:
            out = torch.zeros(st.xres * st.yres * 4, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            seq.submit(st.yres, 0, 1, out.data_ptr(), out.numel(), view=v)
        outs.append((current, out))
    seq.wait()
    for k, (sc, out) in enumerate(outs):
        fresh.set_scene(sc)
        want, _ = fresh.render()
        got = out if isinstance(out, np.ndarray) else out.cpu().numpy().reshape(want.shape)
        frames += 1
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH sequence {it} frame {k}: {(got != want).any(axis=-1).sum()} pixels", flush=True)
    seq.close()
    if it % 10 == 0:
        print(f"sequence {it}: {len(views)} frames of {w0}x{h0}, elapsed {time.time() - t0:.0f}s", flush=True)
fresh.close()
print(f"frames: {frames} mismatches: {bad}")
sys.exit(1 if bad else 0)
