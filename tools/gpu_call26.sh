#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02z
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_campplus.py -m gpu -q -s > $O/pytest_campplus.log 2>&1; echo "pytest_campplus rc=$?" > $O/status.txt
cat $O/status.txt; tail -12 $O/pytest_campplus.log | cut -c1-220
