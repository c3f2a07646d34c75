cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 --durations=6 -k "head_groups or chunk_full_head or test_chunk" > gpurun_out/r02n_pytest.log 2>&1; echo "pytest=$?"; tail -12 gpurun_out/r02n_pytest.log
python - <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
for hh in (4, 8, 16):
    r = bench.measure_chunk(dev, B=64, H=hh, Dk=1024 // hh, Dv=1024 // hh, reps=200)
    print(hh, round(r['ms'], 4), round(r['frac'], 4), round(r['tflops'], 1))
for hh in (8, 16):
    r = bench.measure_chunk(dev, B=8, H=hh, Dk=1024 // hh, Dv=1024 // hh, reps=200)
    print('b8', hh, round(r['ms'], 4), round(r['frac'], 4))
PY
