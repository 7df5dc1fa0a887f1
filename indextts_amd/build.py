"""Build the gfx950 C-ABI library in-tree:  indextts_amd/csrc/libindextts_hip.so

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  Each .hip source is compiled to its own object (in parallel, only when the source or a
header is newer than the object) and the objects are linked into the shared library.
"""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libindextts_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE]
# per-file extras.  VGPR-form MFMA: the accumulators live in ordinary VGPRs (gfx950 MFMA reads / writes either file), so the
# softmax / epilogue VALU code works on them in place -- with AGPR accumulators the flash-attention loop carried 80
# v_accvgpr_read/write moves per key tile and every GEMM epilogue 192.  The BigVGAN conv kernel keeps its tuned allocation.
# No SLP vectorisation in s2mel_kernels.hip / bigvgan_kernels.hip: hipcc otherwise packs adjacent f32 adds / muls / fmas into
# v_pk_*_f32, which cost more than the scalar pair they replace (MI355X guide: ~13 extra cycles each beside MFMAs).  Measured: the
# anti-aliased activation kernel 2.32 -> 2.55 TB/s (profiles/r02l/voc_*.log), the flash-attention solve -4 % (profiles/r02j).
# gpt_kernels.hip keeps SLP vectorisation: building it with -fno-slp-vectorize (tried for the f32x3 operand split) changes how hipcc
# contracts / packs the f32 arithmetic of typical_filter, and a borderline token of the reference-minted `gpt_typical_greedy` fixture
# flipped (profiles/r03j) -- the bit-exact-ids contract outranks a few percent on an optional GEMM mode.
EXTRA = {"gpt_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         "s2mel_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"],
         "bigvgan_kernels.hip": ["-fno-slp-vectorize"],
         "bigvgan_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"],
         "gemm_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]}

def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build() -> bool:
    return _stale(LIB, sources() + headers() + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libindextts_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = headers() + [os.path.abspath(__file__)]

    def compile_one(src):
        obj = _obj(src)
        if not force and not _stale(obj, [src] + hdrs):
            return None
        cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + r.stdout + r.stderr)
        return obj

    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [_obj(s) for s in srcs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    check_isa()
    return LIB


def check_isa() -> None:
    """Fail the build when hipcc has emitted, in a kernel that stages its accumulator tile through LDS (gemm_tile.h's tile store), the in-place
    packed-f32 form behind round 4's run-to-run differences (tools/check_isa.py: rule and history).  The bit-stability of the flow-matching solve
    rests on that form being absent; SLP vectorisation stays on in gpt_kernels.hip (bit-exact sampler fixtures), so the generated code is checked."""
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_isa.py")
    if not os.path.exists(tool):
        return
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("ISA check failed (in-place packed f32 op with a half selection in a tile-store kernel):\n" + r.stdout + r.stderr)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
