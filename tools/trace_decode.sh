#!/bin/bash
# GPU box: kernel-trace one short GPT generate and print the per-kernel timeline of one decode step (durations + gaps).
# usage: tools/trace_decode.sh [B=8] [tokens=40]  (the step shown is the second-to-last: context = 134 + tokens - 2)
set -u
B=${1:-8}
N=${2:-40}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/trace_decode
mkdir -p "$OUT"
cat > /tmp/short_gen.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import gpt, synth
gcfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(spk_cond_mode="campplus", **gcfg, precision="bf16", device="cuda:0")
m.load_state_dict(synth.gpt_weights(gcfg, suppress_eos=True))
m.post_init_gpt2_config(kv_cache=True, half=True)
B = $B
text = torch.randint(2, 12000, (B, 128)).cuda(); langs = torch.full((B,), 3, dtype=torch.long).cuda()
style = torch.randn(1, 192).cuda(); emo = (torch.randn(1, 1280) * 0.1).cuda()
for n in (8, $N):
    codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=n,
                                  do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0)
torch.cuda.synchronize()
print(m.last_timing)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- python /tmp/short_gen.py > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = glob.glob(os.path.join(out, "raw", "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the sample kernel is the last launch of a decode step (it also advances the step counters); take the step before the last
adv = [i for i, n in enumerate(names) if n.startswith("sample_kernel")]
a0, a1 = adv[-3], adv[-2]
step = rows[a0 + 1:a1 + 1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = int(rows[a0]["End_Timestamp"])
tot_dur = tot_gap = 0
lines = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    lines.append(f"{r['Kernel_Name'][:48]:48s} dur {(e - s) / 1000:7.2f} us  gap {(s - prev_end) / 1000:6.2f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')} wg {r.get('Workgroup_Size_X', '?')} vgpr {r.get('VGPR_Count', '?')} lds {r.get('LDS_Block_Size', '?')}")
    tot_dur += e - s; tot_gap += s - prev_end; prev_end = e
open(os.path.join(out, "step_timeline.txt"), "w").write("\n".join(lines) + f"\nTOTAL kernels {len(step)} dur {tot_dur / 1000:.1f} us gaps {tot_gap / 1000:.1f} us\n")
print("\n".join(lines[:18])); print("..."); print("\n".join(lines[-6:]))
print(f"TOTAL kernels {len(step)} dur {tot_dur / 1000:.1f} us gaps {tot_gap / 1000:.1f} us")
PY
rm -rf "$OUT/raw"
tail -2 "$OUT/run.log"
