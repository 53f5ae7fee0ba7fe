"""Count VALU / SALU instructions per basic block of one kernel in hipcc's -S output.
    python tools/count_valu.py dev.s <kernel-name-substring> [first-label last-label]"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and l.rstrip().endswith((":", ")")) or (l.startswith("_Z") and sys.argv[2] in l and ":" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
label, v, s = "entry", 0, 0
tot_v = tot_s = 0
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    t = l.strip()
    if m or t.startswith("; %bb."):
        if v or s:
            print(f"{label:14s} valu {v:4d} salu {s:4d}")
        label, v, s = (m.group(1) if m else t.split()[1]), 0, 0
        continue
    if t.startswith(("v_", "ds_", "global_", "buffer_", "flat_")):
        v += 1; tot_v += 1
    elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop")):
        s += 1; tot_s += 1
print(f"{label:14s} valu {v:4d} salu {s:4d}")
print("total", tot_v, tot_s)
