import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))

SCENES = ROOT / "scenes"
GOLDEN = ROOT / "tests" / "golden"
SMALL_SCENES = [
    "scene_2spheres_ortho_flat", "scene_2spheres_ortho_phong", "scene_2spheres_persp_flat",
    "scene_2spheres_persp_phong", "scene_gtri_ortho_flat", "scene_gtri_ortho_phong",
    "scene_gtri_persp_flat", "scene_gtri_persp_phong",
]
ALL_SCENES = SMALL_SCENES + ["teatime"]


def free_port():
    """A TCP port nothing listens on right now, for a torch.distributed rendezvous on 127.0.0.1 (a
    fixed number could be taken by another job on the machine)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mrt():
    import mrt_pkg
    return mrt_pkg.load()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_ctx(mrt):
    """One context on GPU 0 for the whole session; fails loudly without the library or a GPU."""
    ctx = mrt.Context(device=0)
    yield ctx
    ctx.close()
