#!/bin/bash
# round 3, call 9: does the chip hold its 2.4 GHz engine clock under sustained MFMA load?  rocm-smi samples (sclk, power) every 0.5 s
# while the f32 / f32x3 / bf16 flow-matching solves and the vocoder run.  (The PMC passes of call 8 suggested ~1.4-1.9 GHz.)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03i
mkdir -p $O
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' ')"; sleep 0.5; done ) > $O/smi.log 2>&1 &
SMI=$!
sleep 2
for prec in fp32 fp32x3 bf16; do
  echo "START $prec $(date +%s.%N)" >> $O/marks.log
  timeout 200 python tools/s2mel_bench.py 64 517 1926 6 $prec 2>&1 | tail -1 >> $O/s2mel.log
  echo "END $prec $(date +%s.%N)" >> $O/marks.log
done
kill $SMI
cat $O/s2mel.log; cat $O/marks.log; grep -c . $O/smi.log; head -3 $O/smi.log | cut -c1-600
python3 - $O <<'PY'
import re, sys
o = sys.argv[1]
marks = [l.split() for l in open(o + "/marks.log")]
spans = {}
for m in marks:
    spans.setdefault(m[1], {})[m[0]] = float(m[2])
rows = []
for l in open(o + "/smi.log"):
    m = re.match(r"t=([\d.]+) (.*)", l)
    if not m: continue
    t = float(m.group(1)); nums = re.findall(r"\((\d+)Mhz\)", m.group(2)); pw = re.findall(r",(\d+\.\d+)", m.group(2))
    rows.append((t, nums, pw))
for k, s in spans.items():
    sel = [r for r in rows if s["START"] + 8 <= r[0] <= s["END"] - 0.5]      # skip model construction
    print(k, "samples", len(sel), "sclk-ish MHz lists:", [r[1] for r in sel][-6:], "power:", [r[2] for r in sel][-6:])
PY
