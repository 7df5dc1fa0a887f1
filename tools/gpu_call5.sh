#!/bin/bash
# GPU box: tap-mode GEMM, packed-f32 softmax, VGPR-form MFMA, scheduled tile-GEMM main loop; new pipeline / v2 / checkpoint-dir tests.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_codec.py tests/test_gpu_pipeline.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" > $O/status.txt
timeout 1800 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_s2mel.py --deselect tests/test_gpu_codec.py --deselect tests/test_gpu_pipeline.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
timeout 200 python tools/prefill_bench.py > $O/prefill_bench.log 2>&1
timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" > $O/s2mel_bench.log
timeout 300 python tools/s2mel_bench.py 16 517 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 8 800 1926 5 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/s2mel_kernel_stats.csv
cat $O/status.txt; tail -5 $O/pytest_new.log; tail -4 $O/pytest.log; cat $O/prefill_bench.log | grep TFLOP; cat $O/s2mel_bench.log; head -c 400 $O/bench.json
