#!/bin/bash
# round 3, call 27 (last GPU seconds of the round): the LayerNorm users outside the GPT suites and smoke(), after the ln_row refactor
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03zz
mkdir -p $O
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/status.txt
timeout 110 python -m pytest tests/test_gpu_w2vbert.py tests/test_gpu_ecapa.py tests/test_gpu_campplus.py tests/test_gpu_frontend.py tests/test_gpu_audio.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
cat $O/status.txt; tail -2 $O/smoke.log | cut -c1-200; tail -3 $O/pytest.log | cut -c1-200
