# bench every library variant under spasm.jl_amd/variants/ (built locally with different tuning constants)
cd $GRAFT_REPO_ROOT
for so in spasm.jl_amd/variants/*.so; do
  echo "== $so"
  SPASM_AMD_LIB=$PWD/$so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(' ms/step %.3f solve %.4f scatter %.4f %s' % (d['ms_per_step'], r['round_ms']['solve'], r['round_ms']['scatter'], r['per_class_ms']))
" || exit 1
done
