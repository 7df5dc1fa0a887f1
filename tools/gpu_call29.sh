#!/bin/bash
# prompt-side DSP + codec quantize at real sizes: HIP-event medians and the rocprofv3 kernel table
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r02zk
mkdir -p $O
timeout 120 python tools/prompt_dsp_bench.py > $O/prompt_dsp.json 2> $O/prompt_dsp.err; echo "bench rc=$?" > $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dsp -o p -- python $GRAFT_REPO_ROOT/tools/prompt_dsp_bench.py > $O/prompt_dsp_prof.json 2> $O/prompt_dsp_prof.err
f=$(find /tmp/prof_dsp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prompt_dsp_kernel_stats.csv
cat $O/status.txt; cat $O/prompt_dsp.json; tail -3 $O/prompt_dsp.err; head -12 $O/prompt_dsp_kernel_stats.csv | cut -c1-160
