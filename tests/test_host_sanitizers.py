"""The host-side C++ (JSON reader, scene loader, PNG writer) under AddressSanitizer and
UBSan with truncated, mutated and random inputs (no GPU code is involved)."""
import subprocess

from conftest import ROOT, SCENES


def test_host_code_under_asan_ubsan(tmp_path):
    src = ROOT / "opencl-raytracing_amd" / "csrc"
    exe = tmp_path / "host_fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT / 'include'}", f"-I{src}", str(ROOT / "tests" / "c" / "host_fuzz.cpp"),
           str(src / "json_min.cpp"), str(src / "scene_loader.cpp"), str(src / "png_writer.cpp"),
           "-o", str(exe), "-lz", "-lpthread"]
    subprocess.run(cmd, check=True)
    r = subprocess.run([str(exe), str(SCENES / "scene_gtri_persp_phong.json"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok ")
