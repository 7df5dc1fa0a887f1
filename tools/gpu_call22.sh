#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02v
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cond.py -m gpu -q -s > $O/pytest_cond.log 2>&1; echo "pytest_cond rc=$?" > $O/status.txt
cat $O/status.txt; tail -40 $O/pytest_cond.log | cut -c1-220
