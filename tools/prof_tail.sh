# Dense-finish profile of config 5 at 1/SCALE (default 10): kernel-time split + MFMA utilisation of the int8 GEMM.
#   tools/prof_tail.sh [scale]      -> gpurun_out/tail_prof/summary.txt
# (rocprofv3 segfaults when the profiled python exits after cooperative launches; the traces are complete by then)
SCALE=${1:-10}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tail_prof && rm -rf gpurun_out/tail_prof/trace gpurun_out/tail_prof/pmc
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail_prof/trace -- python3 tools/time_c5.py $SCALE > gpurun_out/tail_prof/run.log 2> gpurun_out/tail_prof/trace.err
timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/tail_prof/pmc -- python3 tools/time_c5.py $SCALE > gpurun_out/tail_prof/run_pmc.log 2> gpurun_out/tail_prof/pmc.err
python3 - $SCALE <<'PY'
import csv, glob, sys, collections
scale = sys.argv[1]
out = [f"config 5 at 1/{scale} (tools/time_c5.py {scale}), rocprofv3 --kernel-trace --stats; then --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES"]
out += [l.rstrip() for l in open('gpurun_out/tail_prof/run.log') if not l.startswith('{')]
f = glob.glob('gpurun_out/tail_prof/trace/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
dense = ('k_panel', 'k_gemm_i8', 'k_trsm_i8', 'k_dense', 'k_inv_table')
tail = sum(float(r['TotalDurationNs']) for r in rows if any(d in r['Name'] for d in dense))
out.append(f"total kernel time {tot/1e9:.3f} s; dense finish (panel + trsm + gemm + emit) {tail/1e9:.3f} s = {100*tail/tot:.1f}%")
for r in rows[:14]:
    out.append(f"  {r['Name'][:80].ljust(80)} calls {r['Calls']:>7} avg {float(r['AverageNs'])/1e3:10.1f} us total {float(r['TotalDurationNs'])/1e9:7.3f} s {100*float(r['TotalDurationNs'])/tot:5.1f}%")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob('gpurun_out/tail_prof/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
out.append("PMC sums per kernel (all dispatches); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs), the gfx94x formula rocprofv3 falls back to")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:8]:
    gui = v.get('GRBM_GUI_ACTIVE', 0)
    util = 100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) if gui else 0
    out.append(f"  {k[:60].ljust(60)} dispatches {n[(k,'GRBM_GUI_ACTIVE')]:>6} GUI_ACTIVE {gui:.3e} MFMA_BUSY {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3e} MFMA_I8 insts {v.get('SQ_INSTS_VALU_MFMA_I8',0):.3e} MfmaUtil {util:5.1f}%")
open('gpurun_out/tail_prof/summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/tail_prof/trace gpurun_out/tail_prof/pmc
