"""The N > 1 path on the CPU: two gloo ranks render interleaved row bands (the
oracle stands in for the GPU), gather to rank 0, reassemble -- the image must
equal the single-process render."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES, free_port
import scene_py


@pytest.mark.parametrize("world,band,yres", [(2, 16, 100), (2, 64, 96), (3, 7, 50)])
def test_gloo_band_gather_roundtrip(oracle, tmp_path, world, band, yres):
    xres = 80
    out = tmp_path / "img.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "tests" / "multirank_worker.py"), "scene_2spheres_persp_phong", str(xres), str(yres),
           str(band), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    img = np.load(out)
    ref = oracle.render(scene_py.load_scene(SCENES / "scene_2spheres_persp_phong.json", xres=xres, yres=yres))
    assert np.array_equal(img, ref["rgba"])
