# timeline of one Schur step of a 1/G shard: bash tools/shard_trace.sh G   -> gpurun_out/shard_trace/timeline_G.txt
G=${1:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/shard_trace && rm -rf gpurun_out/shard_trace/trace
cat > /tmp/one_shard.py <<PY
import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, spasm_jl_amd as S
lib = S._abi.lib(); n = 1_000_000; G = $G
A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
plan = lib.spasm_amd_schur_plan_create_strided(A.data, G - 1, n, G)
lib.spasm_amd_schur_plan_class_timing(plan, 0)
st = torch.cuda.Stream(); sp = C.c_void_p(st.cuda_stream)
for _ in range(6): lib.spasm_amd_schur_plan_run(plan, sp)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shard_trace/trace -- python3 /tmp/one_shard.py > /dev/null 2> gpurun_out/shard_trace/err.log
python3 - $G <<'PY'
import csv, glob, sys
G = sys.argv[1]
f = glob.glob('gpurun_out/shard_trace/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: from the last k_wbuild_reset on (the W build opens the step; plans that keep to the lists start at k_solve_reset)
idx = max([i for i, r in enumerate(rows) if 'k_wbuild_reset' in r['Kernel_Name']] or [i for i, r in enumerate(rows) if 'k_solve_reset' in r['Kernel_Name']])
step = rows[idx:]
t0 = int(step[0]['Start_Timestamp'])
out = []
for r in step:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    out.append(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  q{r['Queue_Id']:>3} s{r['Stream_Id']:>3}  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6}  {n}")
open(f'gpurun_out/shard_trace/timeline_{G}.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/shard_trace/trace
