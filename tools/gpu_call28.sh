#!/bin/bash
# new rows of this session: prompt-audio front end + codec quantize half (and the refactored codec decode)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02zj
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_audio.py tests/test_gpu_codec.py -x -q -s > $O/pytest_new.log 2>&1; echo "pytest rc=$?" > $O/status.txt
cat $O/status.txt; grep -E "max\|d\||passed|failed|Error|error|equal" $O/pytest_new.log | tail -40
