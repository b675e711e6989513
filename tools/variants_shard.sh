# tools/shard_time.py for every library variant under spasm.jl_amd/variants/
cd $GRAFT_REPO_ROOT
for so in spasm.jl_amd/variants/*.so; do
  echo "== $so"
  SPASM_AMD_LIB=$PWD/$so timeout -k 10 200 python tools/shard_time.py 2>/dev/null | grep "G=" | cut -c1-200 || exit 1
done
