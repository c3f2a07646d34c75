cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/train_prof -o tr -- python bench.py --train --steps 4 --warmup 2 > gpurun_out/r02r_train.json 2> gpurun_out/r02r_train.err
db=$(find /tmp/train_prof -name "*results.db" | head -1); python tools/prof_summary.py "$db" gpurun_out/r02r_train_step_kernel_stats.csv > /dev/null; ls -la gpurun_out | head -5
