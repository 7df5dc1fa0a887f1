#!/bin/bash
# round 4, GPU call 6: the full-size parity tests added last (3-beam beam-sample ids vs the reference's, pipeline level at configs[1] /
# configs[4] sizes vs the oracle chain), the bench line in the new default mode (fp32x3, 6 plane products, x3 attention) with the per-rank
# shards, and the rocprofv3 kernel stats of the same command.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r04e
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "beam_sample" > $O/pytest_fullsize_beam.log 2>&1; echo "pytest fullsize beam rc=$?" | tee $O/status.txt
grep -E "GPT 24|passed|failed|Error|row " $O/pytest_fullsize_beam.log | tail -5
timeout 900 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s > $O/pytest_pipeline_fullsize.log 2>&1; echo "pytest pipeline fullsize rc=$?" | tee -a $O/status.txt
grep -E "configs\[|passed|failed|Error" $O/pytest_pipeline_fullsize.log | tail -12
timeout 900 python bench.py --steps 3 --warmup 1 --no-configs > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04e/bench.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "dtype", j["dtype"], "roofline", j["roofline"])
    print("cpu_baseline", j["cpu_baseline"])
    st = j["stages"]
    print({k: v for k, v in st.items() if not isinstance(v, (dict, list))})
    print("by precision", j.get("value_by_s2mel_precision"))
    print({k: v for k, v in st.get("configs", {}).items()})
except Exception as e:
    print("bench json:", repr(e))
PY
tail -3 $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 0 --no-extras > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof rc=$?" | tee -a $O/status.txt
find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/bench_kernel_stats.csv \; 2>/dev/null
rm -rf $O/raw
head -12 $O/bench_kernel_stats.csv | cut -c1-200
