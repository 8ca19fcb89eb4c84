"""Imports the package directory ``opencl-raytracing_amd/`` (whose name is not a
valid Python identifier) under the module name ``opencl_raytracing_amd``."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG_DIR = ROOT / "opencl-raytracing_amd"
NAME = "opencl_raytracing_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, PKG_DIR / "__init__.py",
                                                  submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
