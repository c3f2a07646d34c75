cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ht in 1 0 1 0; do K2_HT=$ht K2_REPS=1500 timeout 120 python tools/perf_k2.py; done
K2_B=8 K2_HT=0 K2_REPS=3000 timeout 120 python tools/perf_k2.py
