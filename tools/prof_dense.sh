# kernel-time split of an echelonize run dominated by the dense finish: tools/prof_dense.sh <name> <python script> [args]
# writes gpurun_out/<name>/stats.txt (top kernels by total time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$1
NAME=$1; shift
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$NAME/trace -- python3 "$@" > gpurun_out/$NAME/run.log 2> gpurun_out/$NAME/trace.err
python3 - $NAME <<'PY'
import csv,glob,sys
name=sys.argv[1]
f=glob.glob(f'gpurun_out/{name}/trace/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=[f"total kernel time {tot/1e9:.3f} s"]
for r in rows[:25]:
    out.append(f"{r['Name'][:90].ljust(90)} calls {r['Calls']:>8} avg {float(r['AverageNs'])/1e3:10.1f} us total {float(r['TotalDurationNs'])/1e9:8.3f} s {100*float(r['TotalDurationNs'])/tot:5.1f}%")
open(f'gpurun_out/{name}/stats.txt','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
tail -8 gpurun_out/$NAME/run.log
rm -rf gpurun_out/$NAME/trace
