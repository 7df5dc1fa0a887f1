#!/bin/bash
# round 4, GPU call 5: the new full-size parity tests (3-beam beam-sample ids vs the reference's, pipeline level at configs[1] / configs[4] sizes vs
# the oracle chain), the s2mel + x3 suites with the determinism test, a short bench line in the new default mode.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04e
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "beam_sample" > $O/pytest_fullsize_beam.log 2>&1; echo "pytest fullsize beam rc=$?" | tee $O/status.txt
grep -E "GPT 24|passed|failed|Error|row " $O/pytest_fullsize_beam.log | tail -5
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py tests/test_gpu_attn_x3.py -x -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
grep -E "bits|passed|failed|Error" $O/pytest_s2mel.log | tail -8
timeout 1500 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s > $O/pytest_pipeline_fullsize.log 2>&1; echo "pytest pipeline fullsize rc=$?" | tee -a $O/status.txt
grep -E "configs\[|passed|failed|Error" $O/pytest_pipeline_fullsize.log | tail -12
timeout 900 python bench.py --steps 2 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 1 > $O/bench_short.json 2> $O/bench_short.log; echo "bench rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04e/bench_short.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "peak", "frac", "ms_per_step")})
    print("s2mel", {k: v for k, v in j["stages"]["s2mel"].items() if "ms_per_step" in k or "tflops" in k})
    print("by precision", j.get("value_by_s2mel_precision"))
    print("gpt", j["stages"]["gpt_decode_ms_per_token"], "bigvgan", j["stages"]["bigvgan_ms_per_step"])
except Exception as e:
    print("bench json:", repr(e))
PY
tail -5 $O/bench_short.log
