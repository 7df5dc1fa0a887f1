#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gpt.py -m gpu -q -x -k "stream or chunk or graph or golden" > $O/pytest_stream.log 2>&1; echo "pytest_stream rc=$?" > $O/status.txt
cat $O/status.txt; tail -30 $O/pytest_stream.log
