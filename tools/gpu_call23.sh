#!/bin/bash
# GPU box: checkpoint -- smoke, full GPU suite, bench, rocprof kernel stats of the bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02y
mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/status.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/status.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
ITTS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --gen-tokens 60 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "bench_dist1 rc=$?" >> $O/status.txt
cat $O/status.txt; tail -2 $O/smoke.log; tail -3 $O/pytest_all.log; head -c 400 $O/bench.json; echo; head -c 200 $O/bench_dist1.json
