#!/bin/bash
# round 6 session G: K1w at ONE workgroup per CU (LDS pad): does the second engine's work run beside it?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in "" tools/abl/liblina_k1wpad32768.so tools/abl/liblina_k1wpad65536.so; do
  echo "== LINA_GLA_LIB=$lib" >> gpurun_out/r06_k1w_pad.txt
  LINA_GLA_LIB=$lib timeout 300 python tools/probe_two_engines.py 256 2 2>&1 | tail -1 >> gpurun_out/r06_k1w_pad.txt
  LINA_GLA_LIB=$lib timeout 300 python tools/perf_loop.py 512 400 2>&1 | tail -1 >> gpurun_out/r06_k1w_pad.txt
  LINA_GLA_LIB=$lib timeout 300 python tools/probe_two_engines.py 128 4 2>&1 | tail -1 >> gpurun_out/r06_k1w_pad.txt
done
cat gpurun_out/r06_k1w_pad.txt
