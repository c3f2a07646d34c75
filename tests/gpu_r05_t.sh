#!/bin/bash
# round 5, session T: DecodeEngineGroup -- its tests, then the bench's two_engines block through a short bench run
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "two_engines" > gpurun_out/r05t_pytest.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r05t_pytest.log
timeout 600 python - <<'PY' 2>/dev/null | tee gpurun_out/r05t_two_engines.json
import json, sys, torch
sys.path.insert(0, ".")
import bench
from lina_speech_amd.configs import l169
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (512, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
print(json.dumps(bench.measure_two_engines(m, texts, dev)))
PY
