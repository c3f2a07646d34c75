cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tests/gpu_k2_prof.sh r02
for lib in "" tools/abl/liblina_wnt_in.so; do
  for r in 1 2; do
  LINA_GLA_LIB=$lib timeout 300 python bench.py --no-train --no-cpu-baseline --no-chunk 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('lib=[$lib]', round(j['value']), round(j['ms_per_step'],4))"
  done
done
