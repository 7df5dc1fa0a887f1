#!/bin/bash
# full validation of the round: every GPU test, smoke(), the default bench line, (kernel stats of the same bench command: profiles/r02zf)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zm
mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/status.txt
tail -4 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
timeout 60 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 120 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd $GRAFT_REPO_ROOT
cat $O/status.txt; cat $O/pytest_gpu_tail.txt; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; tail -3 $O/bench.err | cut -c1-300
