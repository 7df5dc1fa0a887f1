#!/bin/bash
# The exact command lists of round 6's GPU sessions (each ran on an MI355X box through gpurun; outputs under gpurun_out/, the summaries that are
# evidence were copied to profiles/r06*).  usage: tools/gpu_r06_calls.sh <n>      e.g.  gpurun --timeout 1200 -- 'bash tools/gpu_r06_calls.sh 1'
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD

# round 6, GPU call 1: per-slot cache positions / steps in the decode session (admission at any step, one session per call): the admission tests,
# every GPT test (the sampler, the QKV epilogues and the KV-cache attention changed), the in-flight schedule against drained batches.
call1() {
    O=$PWD/gpurun_out/r06a
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_admission.py tests/test_gpu_compaction.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|in-flight schedule|passed|failed|Error|error" $O/pytest_admission.log | tail -12
    timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_gpt.log
    timeout 600 python tools/inflight_bench.py 512 64 32 8 120 560 > $O/inflight_512_120.log 2>&1; echo "inflight rc=$?" | tee -a $O/status.txt; tail -1 $O/inflight_512_120.log
    timeout 600 python tools/inflight_bench.py 512 64 32 8 280 560 > $O/inflight_512_280.log 2>&1; tail -1 $O/inflight_512_280.log
    timeout 600 python tools/inflight_bench.py 128 64 32 8 120 560 > $O/inflight_128_120.log 2>&1; tail -1 $O/inflight_128_120.log
    timeout 600 python tools/inflight_bench.py 512 64 16 4 120 560 > $O/inflight_512_120_c16.log 2>&1; tail -1 $O/inflight_512_120_c16.log
}

"call$1"
