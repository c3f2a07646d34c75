#!/bin/bash
# round 5, session O: tall kernels with 64-row workgroups (LINA_TALL_MTW=1 build) -- parity of the variant build, then timing
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_GLA_LIB=tools/abl/liblina_tall_m1.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05o_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05o_pytest.log
for V in product m1; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for TV in 0 2; do LINA_GLA_LIB=$LIB LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | sed "s/^/$V /"; done
  LINA_GLA_LIB=$LIB LINA_TALL=1 LINA_TALL_V=0 timeout 60 python tools/perf_tall.py 256 40 2>/dev/null | sed "s/^/$V /"
done | tee gpurun_out/r05o_rows.txt
LINA_GLA_LIB=tools/abl/liblina_tall_m1.so timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05o_rows.txt
