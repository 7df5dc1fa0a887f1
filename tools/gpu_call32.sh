#!/bin/bash
# EngineFrontend: prompt side from a checkpoint directory without the reference package
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zn
mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_frontend.py -x -q -s > $O/pytest_frontend.log 2>&1; echo "pytest rc=$?" > $O/status.txt
cat $O/status.txt; grep -E "max\|d\||passed|failed|Error|error|assert" $O/pytest_frontend.log | tail -30
