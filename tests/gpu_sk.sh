#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-sk}
python tools/perf_skinny.py 2>&1 | grep -E "us/launch"
SK=inproj timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d gpurun_out/${TAG}_pmc1 -o ${TAG} --output-format csv -- python tools/perf_skinny.py > gpurun_out/${TAG}_pmc1.log 2>&1; echo "pmc1=$?"
SK=inproj timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B -d gpurun_out/${TAG}_pmc2 -o ${TAG} --output-format csv -- python tools/perf_skinny.py > gpurun_out/${TAG}_pmc2.log 2>&1; echo "pmc2=$?"
SK=inproj timeout 600 rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ -d gpurun_out/${TAG}_pmc3 -o ${TAG} --output-format csv -- python tools/perf_skinny.py > gpurun_out/${TAG}_pmc3.log 2>&1; echo "pmc3=$?"
