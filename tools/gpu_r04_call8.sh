#!/bin/bash
# round 4, GPU call 8: vmcnt retirement order across VGPR loads / LDS-DMA (microbenchmark), the configs[3]-size pipeline parity test, and the
# PMC passes behind the bench line's traffic figures (x3 GEMM, vocoder conv) and the GPT decode / prefill counters at 64 rows.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r04g
mkdir -p $O
timeout 120 tools/microbench/bin/vmcnt_order > $O/vmcnt_order.log 2>&1; echo "vmcnt_order rc=$?" | tee $O/status.txt
cat $O/vmcnt_order.log
timeout 900 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s -k config3 > $O/pytest_config3.log 2>&1; echo "pytest config3 rc=$?" | tee -a $O/status.txt
grep -E "configs\[|passed|failed|Error|error" $O/pytest_config3.log | tail -8
timeout 600 bash tools/pmc_s2mel_traffic.sh 64 fp32x3 > $O/pmc_s2mel.log 2>&1; echo "pmc_s2mel rc=$?" | tee -a $O/status.txt
tail -2 $O/pmc_s2mel.log | cut -c1-600
cp gpurun_out/pmc_s2mel/s2mel_gemm_traffic.json $O/ 2>/dev/null
timeout 900 bash tools/pmc_gpt.sh 64 24 > $O/pmc_gpt.log 2>&1; echo "pmc_gpt b64 rc=$?" | tee -a $O/status.txt
tail -3 $O/pmc_gpt.log | cut -c1-800
cp gpurun_out/pmc_gpt/gpt_pmc_b64.json $O/ 2>/dev/null
timeout 600 bash tools/pmc_bench_traffic.sh 64 > $O/pmc_conv.log 2>&1; echo "pmc_conv rc=$?" | tee -a $O/status.txt
tail -2 $O/pmc_conv.log | cut -c1-600
ls gpurun_out/pmc_bench/ | head; cp gpurun_out/pmc_bench/*.json $O/ 2>/dev/null
