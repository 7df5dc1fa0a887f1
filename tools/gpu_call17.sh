#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_edges.py -m gpu -q -x > $O/pytest_voc.log 2>&1; echo "pytest_voc rc=$?" > $O/status.txt
timeout 300 python tools/voc_stage_profile.py 16 > $O/voc.log 2>&1
cat $O/status.txt; tail -3 $O/pytest_voc.log; head -3 $O/voc.log; grep " act " $O/voc.log | head -20
