#!/bin/bash
# round 4, GPU call 13: the round's full validation -- every GPU test, smoke, the bench line as the driver runs it, the rocprofv3 kernel stats of
# the same command, and the bf16 stage trace with the wqkv fusion off (default) / on.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r04m
mkdir -p $O
timeout 300 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:s2mel_fused=2 > $O/trace_bf16.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
grep -v "amdgpu.ids" $O/trace_bf16.log | grep -v "   repetition" | cut -c1-400
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04m/bench.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "peak", "frac", "traffic", "ms_per_step")})
    print("by precision", j.get("value_by_s2mel_precision"))
    st = j["stages"]
    print({k: v for k, v in st.items() if not isinstance(v, (dict, list))})
    print({k: (v.get("audio_seconds_per_sec"), v.get("ms_per_step")) for k, v in st.get("configs", {}).items() if isinstance(v, dict)})
    print("cpu", j["cpu_baseline"]["value"])
except Exception as e:
    print("bench json:", repr(e))
PY
tail -3 $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 0 --no-extras > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof rc=$?" | tee -a $O/status.txt
find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/bench_kernel_stats.csv \; 2>/dev/null
rm -rf $O/raw
head -8 $O/bench_kernel_stats.csv | cut -c1-160
