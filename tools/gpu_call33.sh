#!/bin/bash
# rocprofv3 kernel stats of one bench step on the round's final code state (pairs with profiles/r02zm/bench.json)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r02zm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/bench_prof.err
echo "prof rc=$?"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
head -6 $O/bench_kernel_stats.csv | cut -c1-150; cut -c1-200 $O/bench_prof.json
