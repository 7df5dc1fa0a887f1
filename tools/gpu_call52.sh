#!/bin/bash
# round 3, call 18: f32 flash attention -- left-over query block split + query cut of the last attention: tests, A/B timings, bit fingerprints
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py -x -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" > $O/status.txt
run_s2() { env "$@" timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed "s/^/[$*] /" >> $O/s2mel_ab.log; }
run_s2 ITTS_NOP=1
run_s2 ITTS_FA32_SPLIT=0
run_s2 ITTS_S2MEL_QCUT=0
run_s2 ITTS_FA32_SPLIT=0 ITTS_S2MEL_QCUT=0
run_s2 ITTS_NOP=1
env timeout 200 python tools/s2mel_bench.py 64 517 1926 5 bf16 2>&1 | tail -1 | sed "s/^/[bf16] /" >> $O/s2mel_ab.log
env ITTS_S2MEL_QCUT=0 timeout 200 python tools/s2mel_bench.py 64 517 1926 5 bf16 2>&1 | tail -1 | sed "s/^/[bf16 QCUT=0] /" >> $O/s2mel_ab.log
cat $O/status.txt; grep -E "passed|failed|rror" $O/pytest_s2mel.log | tail -5 | cut -c1-200; cat $O/s2mel_ab.log | cut -c1-330
