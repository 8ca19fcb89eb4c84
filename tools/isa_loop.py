"""Prints the instruction mix of the hottest basic blocks of one kernel in a device assembly
listing (hipcc --cuda-device-only -S).  Usage: python tools/isa_loop.py file.s <kernel substring> [n]"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith(key) or (l.startswith("_ZN") and key in l.split(":")[0]))
end = next(i for i in range(start, len(txt)) if ".amdhsa_kernel" in txt[i])
blocks, cur, name = [], [], "entry"
for l in txt[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur))
        name, cur = l.split(":")[0], []
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur.append(l.strip().split()[0])
blocks.append((name, cur))
score = lambda b: sum(1 for i in b[1] if i.startswith(("v_fma", "v_mul_f32", "v_fmac", "v_add_f32", "v_sub_f32", "v_min", "v_max")))
for name, ins in sorted(blocks, key=score, reverse=True)[:top]:
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"{name}: {len(ins)} instructions, {valu} VALU")
    print("   ", ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
