#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of one bench step, summaries into gpurun_out/prof_<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o bench -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rocprofv3 exit: $?" >> "$OUT/bench.err"
# keep only the small summaries (kernel stats), not the full trace
find "$OUT/raw" -name "*kernel_stats*.csv" -exec cp {} "$OUT/kernel_stats.csv" \; 2>/dev/null
find "$OUT/raw" -name "*domain_stats*.csv" -exec cp {} "$OUT/domain_stats.csv" \; 2>/dev/null
ls -la "$OUT/raw" > "$OUT/raw_listing.txt" 2>&1
find "$OUT/raw" -type f | head -50 >> "$OUT/raw_listing.txt"
du -sh "$OUT/raw" >> "$OUT/raw_listing.txt"
rm -rf "$OUT/raw"
python3 - "$OUT" <<'PY'
import csv, sys, os, collections
out = sys.argv[1]
p = os.path.join(out, "kernel_stats.csv")
if os.path.exists(p):
    rows = list(csv.DictReader(open(p)))
    print("kernel_stats columns:", list(rows[0].keys()) if rows else None)
    for r in rows[:25]:
        print({k: (v[:70] if isinstance(v, str) else v) for k, v in r.items()})
PY
cat "$OUT/bench.json"
