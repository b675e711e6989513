# more SQ counters per kernel: bash tools/pmc_insts2.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2_$tag
for kv in "$@"; do export "$kv"; done
C="SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES"
timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc2_$tag/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2_$tag/bench1.json 2> gpurun_out/pmc2_$tag/err1.log || echo "pass failed"
