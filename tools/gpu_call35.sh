#!/bin/bash
# First GPU call of the next round (prepared at the end of round 2, not run): the grid-barrier measurement the persistent decode kernel
# depends on, then the full validation (every GPU test, smoke, default bench line).  Build the microbenchmark first:
#   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/bin/grid_barrier tools/microbench/grid_barrier.hip
# (the binary is git-ignored but travels with the gpurun snapshot).  Every spin loop in it is bounded; it still runs under `timeout`.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03a
mkdir -p $O
if [ -x tools/microbench/bin/grid_barrier ]; then
  timeout 60 tools/microbench/bin/grid_barrier 4000 > $O/grid_barrier.log 2>&1; echo "grid_barrier rc=$?" > $O/status.txt
else
  echo "grid_barrier binary missing" > $O/status.txt
fi
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -4 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
timeout 90 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 150 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; cat $O/grid_barrier.log 2>/dev/null; cat $O/pytest_gpu_tail.txt; tail -2 $O/smoke.log; cut -c1-300 $O/bench.json
