# final profile set of the round: kernel trace stats + PMC traffic passes + FETCH_SIZE calibration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/fcal tools/fetch_calib.hip || exit 1
for C in FETCH_SIZE WRITE_SIZE; do timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/final/calib_$C -- /tmp/fcal > gpurun_out/final/calib_$C.log 2>&1; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_trace.json 2> gpurun_out/final/trace.err
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do t=$(echo $C | tr ' ' '_'); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/final/pmc_$t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_$t.json 2> gpurun_out/final/pmc_$t.err; done
timeout -k 10 400 python bench.py > gpurun_out/final/bench_full.json 2> gpurun_out/final/bench_full.err
ls gpurun_out/final
# diagnostic builds (if present): phase ablations and instruction mix of the scatter / combine kernels
if [ -f build/diag/libspasm_amd_ablate.so ]; then
  (export SPASM_AMD_LIB=$PWD/build/diag/libspasm_amd_ablate.so; bash tools/ablate_stream.sh > gpurun_out/final/ablate_scatter.txt 2>&1; bash tools/ablate_combine.sh > gpurun_out/final/ablate_combine.txt 2>&1)
fi
bash tools/pmc_insts.sh final > /dev/null 2>&1
python tools/shard_time.py 2>/dev/null | grep "G=" > gpurun_out/final/shard_time.txt
