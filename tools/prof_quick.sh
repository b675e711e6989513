cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/q1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q1/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q1/bench_trace.json 2> gpurun_out/q1/trace.err
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do t=$(echo $C | tr ' ' '_'); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/q1/pmc_$t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/q1/bench_$t.json 2> gpurun_out/q1/pmc_$t.err; done
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/q1/trace/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
for C in ['FETCH_SIZE','WRITE_SIZE','TCC_HIT_sum_TCC_MISS_sum']:
    acc=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/q1/pmc_{C}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name'].split('(')[0].replace('void ','')
            if n.startswith('k_wstream') or n.startswith('k_wplan') or n.startswith('k_stream_fix') or n.startswith('k_scatter<1'):
                acc[(n[:40],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(k, round(sum(v)/len(v)/1e6,3), 'M')
PY
