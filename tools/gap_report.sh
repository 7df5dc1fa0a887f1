#!/bin/bash
# GPU box: where does the timed step's wall time go that is NOT kernel time?  rocprofv3 kernel trace of `bench.py --steps 1 --warmup 1` (headline leg
# only), then: the last step's window, the sum of its kernel durations, the idle time between consecutive kernels, and the largest idle gaps with the
# kernels on either side.  usage: tools/gap_report.sh <tag>   -> gpurun_out/gap_<tag>/report.txt
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gap_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY' | tee "$OUT/report.txt"
import csv, glob, json, os, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "raw", "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
try:
    ms = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])["ms_per_step"]
except Exception:
    ms = 10300.0
t_end = max(e[1] for e in ev)
# the timed step ends with the int16 conversion of the waveforms; everything after it (if anything) is teardown
win0 = t_end - int(ms * 1e6)
step = [e for e in ev if e[0] >= win0]
busy, idle, gaps, cur_end, prev = 0, 0, [], step[0][0], None
for s, e, n in step:
    if s > cur_end:
        idle += s - cur_end
        gaps.append((s - cur_end, prev, n, cur_end - win0))
    busy += max(0, e - max(s, cur_end)) if e > cur_end else 0
    if e > cur_end:
        cur_end, prev = e, n
print(f"step window {ms:.1f} ms: {len(step)} kernels, device busy {busy / 1e6:.1f} ms, idle between kernels {idle / 1e6:.1f} ms")
short = lambda n: (n or "-").split("(")[0][-60:]
by_pair = {}
for g, a, b, t in gaps:
    k = (short(a), short(b))
    v = by_pair.setdefault(k, [0, 0])
    v[0] += g; v[1] += 1
print("idle time by (kernel before, kernel after), top 25:")
for (a, b), (g, c) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g / 1e6:9.2f} ms in {c:6d} gaps  {a}  ->  {b}")
print("largest single gaps:")
for g, a, b, t in sorted(gaps, key=lambda x: -x[0])[:25]:
    print(f"  {g / 1e6:8.3f} ms at +{t / 1e6:9.1f} ms  {short(a)}  ->  {short(b)}")
PY
rm -rf "$OUT/raw"
