#!/bin/bash
# round 4, GPU call 7: the three-stage A ring of the x3 GEMM (option x3_stages = 3) -- bitwise test against the two-stage kernel, then the
# 8 x 2443-frame solve in alternating pairs, under rocprofv3 so that the per-kernel averages of both variants come from one process.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r04f
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s -k "three_stage" > $O/pytest_ring3.log 2>&1; echo "pytest ring3 rc=$?" | tee $O/status.txt
tail -3 $O/pytest_ring3.log
timeout 300 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 fp32x3:x3_stages=3 fp32x3 fp32x3:x3_stages=3 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
grep "^B=" $O/s2mel_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o sb -- python $ROOT/tools/s2mel_bench.py 8 517 1926 2 fp32x3 fp32x3:x3_stages=3 > $O/s2mel_bench_prof.log 2>&1; echo "rocprof rc=$?" | tee -a $O/status.txt
find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/s2mel_kernel_stats.csv \; 2>/dev/null
rm -rf $O/raw
grep "gemm_x3" $O/s2mel_kernel_stats.csv | cut -c1-160
