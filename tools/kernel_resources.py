"""Build container: register / LDS / occupancy table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py gpt_kernels.hip [name filter ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indextts_amd"))
import build as B  # noqa: E402
src = os.path.join(B.CSRC, sys.argv[1])
filt = sys.argv[2:]
cmd = ["hipcc"] + B.FLAGS + B.EXTRA.get(os.path.basename(src), []) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for blk in re.split(r"remark: Function Name: ", txt)[1:]:
    name = blk.split(" [-Rpass")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    if filt and not any(f in dem for f in filt):
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", blk) or [None, "?"])[1]
    print(f"{dem[:90]:90s} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")
