# timing ablations of k_combine (diagnostic build: make -C spasm.jl_amd/csrc ablate); results are wrong when SPASM_DBG != 0
# 16 = no Uinv loads, 32 = no inserts, 64 = no header gathers, 128 = no record stores
export SPASM_AMD_LIB=$GRAFT_REPO_ROOT/build/diag/libspasm_amd_ablate.so
for d in 0 16 32 48 64 128 192 240; do echo "DBG=$d"; SPASM_DBG=$d timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(' ms/step', round(d['ms_per_step'],3), 'solve', r['round_ms']['solve'], 'scatter', r['round_ms']['scatter'])"; done
