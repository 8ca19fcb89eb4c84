"""include/mrt.h is a C header: a C99 program compiled with gcc checks the struct
layouts and drives the loader through the shared library (no GPU needed)."""
import subprocess

from conftest import ROOT, SCENES


def test_header_compiles_as_c_and_layouts_match(tmp_path):
    exe = tmp_path / "abi_check"
    pkg = ROOT / "opencl-raytracing_amd"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "c" / "abi_check.c"), "-o", str(exe), f"-L{pkg}", "-lmrt_hip",
                    f"-Wl,-rpath,{pkg}"], check=True)
    r = subprocess.run([str(exe), str(SCENES / "teatime.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout == "400 400 2 4032\nok\n"
