"""Code-object metadata of every kernel in the library (registers, LDS, scratch, spills), taken
from the device assembly hipcc emits for each .hip source with the Makefile's flags.

Usage: python tools/kernel_meta.py profiles/r2/kernel_metadata.csv

rocprofv3's VGPR_Count column on gfx950 reports half of .vgpr_count (80 -> 40, 120 -> 60): the
numbers here are the ones the occupancy rule applies to (512 VGPRs per SIMD lane, granule 8).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "opencl-raytracing_amd"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         f"-I{ROOT / 'include'}", f"-I{PKG / 'csrc'}", "--cuda-device-only", "-S"]
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
        "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "wavefront_size")


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def main(out):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in sorted((PKG / "csrc").glob("*.hip")):
            asm = Path(d) / (src.stem + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(src), "-o", str(asm)], check=True, capture_output=True)
            text = asm.read_text()
            m = re.search(r"amdhsa\.kernels:(.*?)amdhsa\.target", text, re.S)
            if not m:
                continue
            for block in re.split(r"\n  - ", m.group(1))[1:]:
                f = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
                if "name" in f:
                    rows.append((src.name, f["name"], [f.get(k, "") for k in KEYS]))
    names = demangle([r[1] for r in rows])
    with open(out, "w") as f:
        f.write("source,kernel," + ",".join(KEYS) + ",waves_per_simd_by_vgprs\n")
        for (src, _, vals), name in zip(rows, names):
            v = int(vals[0] or 0) + int(vals[1] or 0)
            occ = min(8, 512 // max(8, (v + 7) // 8 * 8)) if v else 8
            f.write(f'{src},"{name}",' + ",".join(vals) + f",{occ}\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kernel_metadata.csv")
