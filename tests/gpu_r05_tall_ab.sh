#!/bin/bash
# round 5: the A/B sessions of the tall projection kernels (linear_tall.h), one short GPU session per phase, in the order they were
# run; their outputs were appended to profiles/r05_tall_perf.txt / r05_tall_pmc.txt / r05_tall_sq.json.  Variant libraries:
# tools/tall_variants.sh.    bash tests/gpu_r05_tall_ab.sh <phase>
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
case "$1" in
c)  # round 5, session C: the tall projection kernels -- parity, then the decode loop at B = 512 / 256 / 128 with and without them
TAG=${1:-r05c}
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall or linear_skinny or inproj" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest=$?"; tail -5 gpurun_out/${TAG}_pytest.log
rm -f gpurun_out/${TAG}_loop.txt
for BB in 512 256 128; do
  LINA_TALL=0 timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/${TAG}_loop.err | tee -a gpurun_out/${TAG}_loop.txt
  timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/${TAG}_loop.err | tee -a gpurun_out/${TAG}_loop.txt
done
rm -rf /tmp/kp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/${TAG}_b512_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_b512_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/${TAG}_b512_step_timeline.csv > gpurun_out/${TAG}_b512_step_timeline.txt; head -8 gpurun_out/${TAG}_b512_step_timeline.txt; tail -3 gpurun_out/${TAG}_b512_step_timeline.txt
;;
d)  # round 5, session D: LDS ring shapes of the tall projection kernels (tools/tall_variants.sh) in the decode loop at B = 512 / 256
rm -f gpurun_out/r05d_variants.txt
for V in product 4x2 6x1 6x2 10x1 12x1; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for BB in 512 256; do
    echo -n "$V: " >> gpurun_out/r05d_variants.txt
    LINA_GLA_LIB=$LIB timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/r05d.err >> gpurun_out/r05d_variants.txt
  done
done
cat gpurun_out/r05d_variants.txt
for V in 6x2 12x1; do
rm -rf /tmp/kp; LINA_GLA_LIB=tools/abl/liblina_tall_$V.so timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/r05d_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r05d_${V}_step_timeline.csv > gpurun_out/r05d_${V}_step_timeline.txt; head -6 gpurun_out/r05d_${V}_step_timeline.txt; tail -2 gpurun_out/r05d_${V}_step_timeline.txt
done
;;
e)  # round 5, session E: the register-ring variant of the tall projection kernels (LINA_TALL_V=1) vs the LDS ring (0) vs the 64-row kernels
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05e_pytest.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r05e_pytest.log
rm -f gpurun_out/r05e_loop.txt
for BB in 512 256 128; do
  for V in 0 1; do
    echo -n "V=$V " >> gpurun_out/r05e_loop.txt
    LINA_TALL_V=$V timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/r05e.err >> gpurun_out/r05e_loop.txt
  done
done
cat gpurun_out/r05e_loop.txt
rm -rf /tmp/kp; LINA_TALL_V=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/r05e_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r05e_v1_step_timeline.csv > gpurun_out/r05e_v1_step_timeline.txt; head -6 gpurun_out/r05e_v1_step_timeline.txt; tail -2 gpurun_out/r05e_v1_step_timeline.txt
;;
f)  # round 5, session F: HBM-side bytes (FETCH_SIZE) and L2 hit / miss of the tall projection kernels, micro script (few launches)
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05f_perf_tall.txt
LINA_TALL=0 timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | tee -a gpurun_out/r05f_perf_tall.txt
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf /tmp/pm; LINA_TALL_V=0 timeout 120 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm --output-format csv -- python tools/perf_tall.py 512 4 > gpurun_out/r05f_$T.log 2>&1; echo "$T=$?"
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee gpurun_out/r05f_pmc_$T.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:8]:
    print(k.ljust(72), "  ".join(f"{c}: n={len(v)} mean={sum(v)/len(v):.1f}" for c, v in d.items()))
PY
done
;;
g)  # round 5, session G: tall kernels with the XCD-aware tile order -- micro timing, FETCH_SIZE, then the decode loop
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05g_perf_tall.txt
for MM in 256 128; do for TL in 0 1; do LINA_TALL=$TL timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done; done | tee -a gpurun_out/r05g_perf_tall.txt
rm -rf /tmp/pm; LINA_TALL_V=0 timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm -o pm --output-format csv -- python tools/perf_tall.py 512 4 > gpurun_out/r05g_FETCH.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r05g_pmc_FETCH_SIZE.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:4]:
    print(k.ljust(72), "  ".join(f"{c}: n={len(v)} mean={sum(v)/len(v):.1f}" for c, v in d.items()))
PY
for BB in 512; do timeout 300 python tools/perf_loop.py $BB 2>/dev/null | tee -a gpurun_out/r05g_perf_tall.txt; done
;;
k)  # round 5, session K: register-ring depth of the tall kernels' variant 1 (D = 4 product, 6, 8)
for V in product d6 d8; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  LINA_GLA_LIB=$LIB LINA_TALL_V=1 timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | sed "s/^/$V /"
done | tee gpurun_out/r05k_depth.txt
;;
l)  # round 5, session L: SQ counters of the tall projection kernels (variant 0 and 1), micro script
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for TV in 0 1; do
  i=0
  for SET in "$SQA" "$SQB"; do
    i=$((i+1))
    rm -rf /tmp/sqt_${TV}_$i; LINA_TALL_V=$TV timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sqt_${TV}_$i -o t --output-format csv -- python tools/perf_tall.py 512 4 > /dev/null 2>&1; echo "sq tall v$TV $i=$?"
  done
  python tools/pmc_sq.py gpurun_out/r05l_tall_v${TV}_sq.json "inproj|gla_inproj_tall_kernel" "up|linear_tall_kernel<unsigned short, true, true" "head|linear_tall_kernel<unsigned short, false, false" -- /tmp/sqt_${TV}_1 /tmp/sqt_${TV}_2
done
;;
m)  # round 5, session M: tall kernels after the loop restructuring (no accumulator copies, statistics on the matrix pipe) -- parity, micro timing, loop
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05m_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05m_pytest.log
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05m_perf_tall.txt
for MM in 256 384; do LINA_TALL=1 LINA_TALL_V=0 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; LINA_TALL=0 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done | tee -a gpurun_out/r05m_perf_tall.txt
timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05m_perf_tall.txt
;;
n)  # round 5, session N: variant 2 of the tall kernels (weights through the LDS ring, A through a register ring)
LINA_TALL_V=2 timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05n_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05n_pytest.log
for TV in 0 2; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05n_perf_tall.txt
for MM in 256; do LINA_TALL=1 LINA_TALL_V=2 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done | tee -a gpurun_out/r05n_perf_tall.txt
LINA_TALL_V=2 timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05n_perf_tall.txt
;;
o)  # round 5, session O: tall kernels with 64-row workgroups (LINA_TALL_MTW=1 build) -- parity of the variant build, then timing
LINA_GLA_LIB=tools/abl/liblina_tall_m1.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05o_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05o_pytest.log
for V in product m1; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for TV in 0 2; do LINA_GLA_LIB=$LIB LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | sed "s/^/$V /"; done
  LINA_GLA_LIB=$LIB LINA_TALL=1 LINA_TALL_V=0 timeout 60 python tools/perf_tall.py 256 40 2>/dev/null | sed "s/^/$V /"
done | tee gpurun_out/r05o_rows.txt
LINA_GLA_LIB=tools/abl/liblina_tall_m1.so timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05o_rows.txt
;;
p)  # round 5, session P: row threshold of the tall kernels with 64-row workgroups: tall vs 64-row split-K kernels at M = 128 .. 512
for MM in 128 192 256 384 512; do for TL in 0 1; do LINA_TALL=$TL timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done; done | tee gpurun_out/r05p_threshold.txt
;;
q)  # round 5, session Q: tall parity on the final thresholds; decode loop at B = 512 / 256 / 192 / 128; K1w state window 8 vs 16 at B = 512 / 256
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall or config4" > gpurun_out/r05q_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05q_pytest.log
for BB in 512 256 192 128; do timeout 300 python tools/perf_loop.py $BB 400 2>/dev/null; done | tee gpurun_out/r05q_loop.txt
for BB in 512 256; do timeout 300 python tools/perf_loop.py $BB 400 16 2>/dev/null; done | tee -a gpurun_out/r05q_loop.txt
for BB in 512; do timeout 300 python tools/perf_loop.py $BB 400 4 2>/dev/null; done | tee -a gpurun_out/r05q_loop.txt
;;
r)  # round 5, session R: stage size of the tall kernels' LDS ring with 64-row workgroups (3 x 2 product vs 3 x 4)
for V in product s31; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for MM in 512 256; do LINA_GLA_LIB=$LIB timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null | sed "s/^/$V /"; done
done | tee gpurun_out/r05r_stage.txt
;;
*) echo "phases: c d e f g k l m n o p q r"; exit 1 ;;
esac
