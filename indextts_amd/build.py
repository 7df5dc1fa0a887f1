"""Build the gfx950 C-ABI library in-tree:  indextts_amd/csrc/libindextts_hip.so

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  Rebuilds only when a source is newer than the library.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libindextts_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libindextts_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", INCLUDE, "-o", LIB + ".tmp"]
    cmd += sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
