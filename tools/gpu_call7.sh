#!/bin/bash
# GPU box: flash attention (double-buffered LDS, permlane max, pinned Q loads), decode-graph cache, per-kernel A/B of the two tile GEMMs.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest_s2mel rc=$?" > $O/status.txt
timeout 600 python -m pytest tests/test_gpu_gpt.py -m gpu -q -s -k "graph or golden or typical" > $O/pytest_gpt_part.log 2>&1; echo "pytest_gpt_part rc=$?" >> $O/status.txt
for qs in 1 2; do
  ITTS_FA_QS=$qs timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/QS=$qs /" >> $O/s2mel_bench.log
done
timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  ITTS_TILE256=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel_$v -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 32 517 1926 3 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof_$v.log 2>&1
  f=$(find /tmp/prof_s2mel_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/s2mel_kernel_stats_b32_tile256_$v.csv
done
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/status.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; tail -3 $O/pytest_s2mel.log; tail -3 $O/pytest_gpt_part.log; tail -3 $O/pytest_all.log; cat $O/s2mel_bench.log; head -c 300 $O/bench.json
