# instruction mix per kernel: bash tools/pmc_insts.sh <tag> [env...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_$tag
for kv in "$@"; do export "$kv"; done
for i in 1 2; do
  case $i in
    1) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES";;
    2) C="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag/bench$i.json 2> gpurun_out/pmc_$tag/err$i.log || echo "pass $i failed"
done
