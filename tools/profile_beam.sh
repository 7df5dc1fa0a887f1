#!/bin/bash
# GPU box: rocprofv3 kernel stats of the 3-beam beam-sample decode at the bench shape (tools/beam_bench.py, first two modes).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_beam
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ITTS_BEAM_BENCH_MODES=2 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o b -- python "$ROOT/tools/beam_bench.py" 64 60 > "$OUT/run.log" 2>&1
find "$OUT/raw" -name "*kernel_stats*.csv" -exec cp {} "$OUT/kernel_stats.csv" \; 2>/dev/null
rm -rf "$OUT/raw"
head -14 "$OUT/kernel_stats.csv" | cut -c1-150
grep "num_beams" "$OUT/run.log" | cut -c1-200
