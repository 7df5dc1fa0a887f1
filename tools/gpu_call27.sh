#!/bin/bash
# full validation of the round: every GPU test, smoke(), the default bench line, and a rocprofv3 kernel-stats pass of one bench step
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zf
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/status.txt
tail -4 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/status.txt; cat $O/pytest_gpu_tail.txt; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; tail -3 $O/bench.err | cut -c1-300
