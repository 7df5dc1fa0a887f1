#!/bin/bash
# experiment: cross-step overlap of the GPT decode (second high-priority stream, worker thread) with the codec / CFM / vocoder of the previous step
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zl
mkdir -p $O
timeout 200 python bench.py --steps 3 --warmup 1 --overlap --no-cpu-baseline --no-extras > $O/bench_overlap.json 2> $O/bench_overlap.err; echo "overlap rc=$?" > $O/status.txt
cat $O/status.txt; grep -E "overlap|warmup|Error|error" $O/bench_overlap.err | tail -20; python - <<'PY'
import json
j=json.load(open("gpurun_out/r02zl/bench_overlap.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], {k:v for k,v in j["stages"].items() if not isinstance(v,dict)})
print(j["stages"]["s2mel"])
PY
