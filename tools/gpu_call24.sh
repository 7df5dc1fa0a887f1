#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02x
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gpt.py -m gpu -q -k "s2mel or tile or prefill or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 32 517 1926 3 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/s2mel_kernel_stats_b32.csv
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -2 $O/pytest.log; cat $O/s2mel_bench.log; head -9 $O/s2mel_kernel_stats_b32.csv | cut -c1-150
