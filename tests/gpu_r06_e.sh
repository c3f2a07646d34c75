#!/bin/bash
# round 6 session E: the deeper token-parity tests and the ADVICE r05 regression test
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r06e timeout 1700 python -m pytest tests/test_model_gpu.py -q -x -s -k "fp32_b64_generate_batch_64 or long_horizon_512 or outputs_are_fresh or hidden" > gpurun_out/r06_e_pytest.log 2>&1; echo "pytest=$?"; grep -v "amdgpu.ids" gpurun_out/r06_e_pytest.log | tail -30
