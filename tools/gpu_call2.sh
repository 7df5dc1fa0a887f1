#!/bin/bash
# GPU box: round-2 second call -- s2mel parity tests first (new code), then the whole GPU suite, s2mel timing + kernel breakdown,
# decode timeline.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest_s2mel rc=$?" > $O/status.txt
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_s2mel.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
timeout 300 python tools/s2mel_bench.py 2 800 1926 25 bf16 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench2 rc=$?" >> $O/status.txt
timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 >> $O/s2mel_bench.log 2>&1; echo "s2mel_bench8 rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 4 800 1926 5 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/s2mel_kernel_stats.csv
timeout 300 bash tools/trace_decode.sh 64 > $O/trace64.log 2>&1; cp gpurun_out/trace_decode/step_timeline.txt $O/step_timeline_b64.txt 2>/dev/null
cat $O/status.txt; tail -5 $O/pytest_s2mel.log; tail -5 $O/pytest.log; cat $O/s2mel_bench.log | tail -4
