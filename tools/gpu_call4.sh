#!/bin/bash
# GPU box: vectorised tile-GEMM epilogue + flash attention variants: whole GPU suite, s2mel timing per QS, end-to-end bench line.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02d
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
for qs in 1 2 4; do
  ITTS_FA_QS=$qs timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/QS=$qs /" >> $O/s2mel_bench.log
done
timeout 300 python tools/s2mel_bench.py 16 517 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 8 800 1926 5 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/s2mel_kernel_stats.csv
cat $O/status.txt; tail -6 $O/pytest.log; cat $O/s2mel_bench.log; tail -3 $O/bench.err; head -c 600 $O/bench.json
