#!/bin/bash
# round 6 session B: main-loop variants of the tall projection kernels (tools/micro/tall_gemm.hip)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for shape in "512 4096 1024" "512 4112 1024" "512 2752 1024" "512 1024 1024"; do
  echo "=== tall_gemm_a $shape" >> gpurun_out/r06_tall_gemm2.txt
  timeout 300 tools/micro/tall_gemm_a $shape >> gpurun_out/r06_tall_gemm2.txt 2>&1
done
cat gpurun_out/r06_tall_gemm2.txt
