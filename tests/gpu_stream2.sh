cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  LINA_GLA_LIB=$1 LINA_DECODE_STREAM=$2 timeout 300 python bench.py --no-train --no-cpu-baseline --no-chunk 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('lib=[$1] stream=[$2]', round(j['value']), round(j['ms_per_step'],4))"
}
for i in 1 2; do run "" none; run tools/abl/liblina_wnt_in.so none; run tools/abl/liblina_wnt.so none; done
