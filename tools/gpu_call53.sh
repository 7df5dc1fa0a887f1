#!/bin/bash
# round 3, call 19: the full validation on the final code state -- every GPU test, smoke, the default bench line,
# rocprofv3 kernel stats of one bench step, PMC traffic of the dominant kernel (s2mel f32 GEMMs), PMC passes of the GPT kernels.
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/r03s
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/status.txt
tail -4 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03s -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof bench rc=$?" >> $O/status.txt
f=$(find /tmp/prof_r03s -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
timeout 400 bash $R/tools/pmc_s2mel_traffic.sh 64 > $O/pmc_s2mel.log 2>&1; echo "pmc s2mel rc=$?" >> $O/status.txt
cp $R/gpurun_out/pmc_s2mel/s2mel_gemm_traffic.json $O/ 2>/dev/null
timeout 400 bash $R/tools/pmc_gpt.sh 24 > $O/pmc_gpt.log 2>&1; echo "pmc gpt rc=$?" >> $O/status.txt
cp $R/gpurun_out/pmc_gpt/gpt_pmc.json $O/ 2>/dev/null
cd $R
cat $O/status.txt; cat $O/pytest_gpu_tail.txt; tail -2 $O/smoke.log; tail -3 $O/bench.err | cut -c1-300; cut -c1-300 $O/bench.json; head -12 $O/bench_kernel_stats.csv | cut -c1-200; tail -2 $O/pmc_s2mel.log | cut -c1-600; tail -5 $O/pmc_gpt.log | cut -c1-300
