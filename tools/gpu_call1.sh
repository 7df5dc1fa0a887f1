#!/bin/bash
# GPU box: round-2 first validation -- all GPU tests, a short bench line, decode timeline, weight-stream policy A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
timeout 300 bash tools/trace_decode.sh 64 > $O/trace64.log 2>&1; echo "trace rc=$?" >> $O/status.txt
cp gpurun_out/trace_decode/step_timeline.txt $O/step_timeline_b64.txt 2>/dev/null
for wnt in 1 0; do
  ITTS_DECODE_WNT=$wnt ITTS_BEAM_BENCH_MODES=1 timeout 300 python tools/beam_bench.py 64 200 > $O/decode_wnt$wnt.log 2>&1
  ITTS_DECODE_WNT=$wnt ITTS_BEAM_BENCH_MODES=1 timeout 300 python tools/beam_bench.py 8 200 >> $O/decode_wnt$wnt.log 2>&1
done
tail -3 $O/pytest.log; cat $O/status.txt; cat $O/decode_wnt*.log | grep "ms/token"
