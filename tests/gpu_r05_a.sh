#!/bin/bash
# round 5, session A: the new generate_batch loop -- kernel + model tests, then end-to-end vs the bare loop at B=64 and B=512
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "pick or cross or softmax_pe or generate_batch or lina_forward or fused_engine or l169_bf16_engine or sampling_loop or config4" > gpurun_out/r05a_pytest.log 2>&1; echo "pytest=$?"; tail -5 gpurun_out/r05a_pytest.log
timeout 600 python tools/perf_generate_batch.py 64 > gpurun_out/r05a_gen64.json 2> gpurun_out/r05a_gen64.err; echo "gen64=$?"; cat gpurun_out/r05a_gen64.json; tail -3 gpurun_out/r05a_gen64.err
timeout 600 python tools/perf_generate_batch.py 512 > gpurun_out/r05a_gen512.json 2> gpurun_out/r05a_gen512.err; echo "gen512=$?"; cat gpurun_out/r05a_gen512.json; tail -3 gpurun_out/r05a_gen512.err
