"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>/) into the files that
are committed under profiles/<round>/: the rocprofv3 kernel-stats CSV as is, plus the mean per
launch of every hardware counter for the render kernels (CSV and JSON).

Usage: python tools/summarize_profile.py gpurun_out/prof_r1b profiles/r1 [name]
name (default teatime4k) prefixes the files written: the same layout comes out of tools/profile.sh
(the 4K benchmark command), tools/profile_stages.sh (name stage_intersect) and, per frame, of
tools/profile_frames.sh (names soup16k, teatime8k).
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path


def main(src, dst, name="teatime4k"):
    src, dst = Path(src), Path(dst)
    dst.mkdir(parents=True, exist_ok=True)
    shutil.copy(src / "trace" / "trace_kernel_stats.csv", dst / f"{name}_kernel_stats.csv")
    accel = src / "trace_accel" / "trace_kernel_stats.csv"
    if accel.is_file():            # tools/time_accel.py teatime4k soup16k: the mrt_accel_* kernels
        shutil.copy(accel, dst / "accel_kernel_stats.csv")
    out = {}
    for d in sorted(src.glob("pmc_*")):
        f = d / "pmc_counter_collection.csv"
        if not f.is_file():
            continue
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mrt::" not in k:
                continue
            # one entry per launch shape: the same kernel runs with different grids (a frame of a
            # context and a frame of a sequence slot are cut into different numbers of slices)
            k = f"{k} [grid {r['Grid_Size']}]"
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = {x: r[x] for x in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
        for (k, c), v in sorted(agg.items()):
            e = out.setdefault(k, {"dispatch": meta[k], "counters_mean_per_launch": {}, "launches": len(v)})
            e["counters_mean_per_launch"][c] = sum(v) / len(v)
    (dst / f"{name}_pmc_summary.json").write_text(json.dumps(out, indent=1))
    with open(dst / f"{name}_pmc_summary.csv", "w") as f:
        f.write("kernel,counter,mean_per_launch\n")
        for k, d in out.items():
            for c, v in d["counters_mean_per_launch"].items():
                f.write(f"\"{k}\",{c},{v:.6g}\n")
    print((dst / f"{name}_kernel_stats.csv").read_text())
    for k, d in out.items():
        if "trace_kernel" in k or "general_intersect" in k:
            print(k, d["dispatch"])
            for c, v in d["counters_mean_per_launch"].items():
                print(f"  {c:28s} {v:.6g}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
