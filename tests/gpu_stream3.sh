cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  LINA_DECODE_STREAM=$1 timeout 300 python bench.py --no-train --no-cpu-baseline --no-chunk 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('stream=[$1]', round(j['value']), round(j['ms_per_step'],4))"
}
for s in "in,up" up "in,up,o" "in,up,head" "up,o" "in,up,o,head" "in,up" none; do run "$s"; done
