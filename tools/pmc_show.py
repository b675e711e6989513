"""print the mean per-dispatch PMC values of tools/pmc_insts.sh for kernels matching a prefix:
   python tools/pmc_show.py gpurun_out/pmc_<tag> 'k_scatter<11' [rows]"""
import csv, glob, collections, sys
d, prefix = sys.argv[1], sys.argv[2]
rows = float(sys.argv[3]) if len(sys.argv) > 3 else None
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n.startswith(prefix):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print(f"  {c:24s} {m:16.0f}" + (f"  {m / rows:10.1f} /row" if rows else ""))
