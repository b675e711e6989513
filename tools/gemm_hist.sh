# per-dispatch durations of k_gemm_i8 grouped by grid size: tools/gemm_hist.sh [scale]
SCALE=${1:-10}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/gemm_hist && rm -rf gpurun_out/gemm_hist/trace
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm_hist/trace -- python3 tools/time_c5.py $SCALE > gpurun_out/gemm_hist/run.log 2> gpurun_out/gemm_hist/trace.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/gemm_hist/trace/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
for r in rows:
    if 'k_gemm_i8' not in r['Kernel_Name']: continue
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']) // 256
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    b = 1
    while b < g: b *= 4
    acc[b][0] += 1; acc[b][1] += d
out = []
for b in sorted(acc):
    out.append(f"grid <= {b:>9} workgroups: {acc[b][0]:>5} dispatches, {acc[b][1]/1e3:9.2f} ms total, {acc[b][1]/acc[b][0]:9.1f} us avg")
open('gpurun_out/gemm_hist/hist.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/gemm_hist/trace
