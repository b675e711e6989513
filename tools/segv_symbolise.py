"""python tools/segv_symbolise.py gpurun_out/segv/trace_coop.err gpurun_out/segv/maps_coop.txt -> library + offset + nearest
dynamic symbol of every frame of the glog backtrace (the libraries of the image are the same here and on the GPU box)."""
import bisect
import re
import subprocess
import sys

err, maps = sys.argv[1], sys.argv[2]
regions = []
for line in open(maps):
    f = line.split()
    if len(f) < 6:
        continue
    lo, hi = (int(x, 16) for x in f[0].split("-"))
    regions.append((lo, hi, int(f[2], 16), f[5]))
syms = {}


def nearest(lib, off):
    if lib not in syms:
        out = subprocess.run(["nm", "-D", "--defined-only", "-C", lib], capture_output=True, text=True).stdout
        out += subprocess.run(["nm", "--defined-only", "-C", lib], capture_output=True, text=True).stdout
        tab = sorted({(int(l.split()[0], 16), " ".join(l.split()[2:])) for l in out.splitlines() if re.match(r"^[0-9a-f]+ [TtWw] ", l)})
        syms[lib] = tab
    tab = syms[lib]
    i = bisect.bisect_right(tab, (off, "\xff")) - 1
    return f"{tab[i][1]} + {off - tab[i][0]:#x}" if i >= 0 else "?"


for line in open(err):
    mm = re.search(r"@\s+0x([0-9a-f]+)", line)
    if not mm:
        continue
    a = int(mm.group(1), 16)
    for lo, hi, fo, path in regions:
        if lo <= a < hi:
            # file offset of the address; for shared objects whose first segment maps at file offset 0 this is the symbol value
            base = min(l for l, h, o, p in regions if p == path)
            print(f"{a:#x}  {path}  +{a - base:#x}  {nearest(path, a - base)}")
            break
    else:
        print(f"{a:#x}  (no mapping)")
