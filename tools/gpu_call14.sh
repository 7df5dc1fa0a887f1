#!/bin/bash
# GPU box: checkpoint -- full GPU suite, bench, rocprof kernel stats of the bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02n
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" > $O/status.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -3 $O/pytest_all.log; head -c 600 $O/bench.json
