# ISA of one kernel and every memory instruction / wait of its main loop, in order -- how the "prefetch under an if" stalls were
# found:  bash tools/isa_waits.sh '_Z9k_scatterILi11ELi128ELi2ELi4ELb1ELi4EEv11ScatterArgs'
# (mangled names: make -C spasm.jl_amd/csrc resources | grep 'Function Name')
sym=${1:?mangled kernel name}
out=${2:-/tmp/isa}
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fopenmp -w -S --cuda-device-only -o $out/engine.s "$(dirname "$0")/../spasm.jl_amd/csrc/engine.hip" || exit 1
awk -v s="^$sym:" '$0 ~ s {p = 1} p {print} p && /^\.Lfunc_end/ {exit}' $out/engine.s > $out/kernel.s
echo "$(wc -l < $out/kernel.s) lines in $out/kernel.s"
grep -n "s_waitcnt vmcnt\|global_load\|global_store\|global_atomic\|s_barrier\|Loop Header" $out/kernel.s
