#!/bin/bash
# ECAPA-TDNN speaker encoder of the v1 / v1.5 vocoder on the engine (one shot: 1.2 GPU-minutes were left)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zo
mkdir -p $O
timeout 45 python -m pytest tests/test_gpu_ecapa.py -q -s > $O/pytest_ecapa.log 2>&1; echo "pytest rc=$?" > $O/status.txt
cat $O/status.txt; grep -E "max\|d\||passed|failed|Error|error|assert|^E " $O/pytest_ecapa.log | tail -30
