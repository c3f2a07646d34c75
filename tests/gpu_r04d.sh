#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r04d timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout=600 -k "config4 or config5" > gpurun_out/r04d_pytest.log 2>&1; echo "pytest=$?"; tail -12 gpurun_out/r04d_pytest.log
