#!/bin/bash
# GPU box: fused s2mel epilogues (parity + A/B timing + kernel breakdown), codec tests, the bf16 GPT gates that failed in r02b.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_codec.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" > $O/status.txt
ITTS_S2MEL_FUSED=0 timeout 300 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s -k "bf16 or production" > $O/pytest_unfused.log 2>&1; echo "pytest_unfused rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_gpt.py -m gpu -q -s -k "bf16 or odd_kblock or typical_sampling" > $O/pytest_gpt.log 2>&1; echo "pytest_gpt rc=$?" >> $O/status.txt
timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 > $O/s2mel_bench.log 2>&1
ITTS_S2MEL_FUSED=0 timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 >> $O/s2mel_bench.log 2>&1
timeout 300 python tools/s2mel_bench.py 16 800 1926 25 bf16 >> $O/s2mel_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 8 800 1926 5 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/s2mel_kernel_stats.csv
cat $O/status.txt; tail -4 $O/pytest_new.log; tail -3 $O/pytest_unfused.log; tail -3 $O/pytest_gpt.log; grep "ms total" $O/s2mel_bench.log
