# bash tools/segv_probe.sh  (GPU box): the profiled run with and without the cooperative panel kernel; raw backtrace + maps under gpurun_out/segv/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/segv && rm -rf gpurun_out/segv/trace
for mode in coop nocoop; do
  export SPASM_PROBE_TAG=$mode
  if [ $mode = nocoop ]; then export SPASM_PROBE_COOP=0; else export SPASM_PROBE_COOP=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/segv/trace -- python3 tools/segv_probe.py > gpurun_out/segv/out_$mode.txt 2> gpurun_out/segv/trace_$mode.err
  echo "$mode: exit code $?" | tee -a gpurun_out/segv/out_$mode.txt
done
rm -rf gpurun_out/segv/trace
grep -c SIGSEGV gpurun_out/segv/trace_coop.err gpurun_out/segv/trace_nocoop.err
