cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in ""; do
  rm -rf /tmp/kp; LINA_GLA_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d /tmp/kp -o kp -- python bench.py --steps 128 --warmup 16 --no-train --no-cpu-baseline --no-chunk > /dev/null 2>&1
  echo "lib=[$lib]"; python tools/prof_positions.py $(find /tmp/kp -name "*results.db" | head -1)
done
