import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch
from indextts_amd import _lib
L = _lib.lib()
for prec, name in ((0, "f32"), (1, "bf16"), (2, "f32x3")):
    n = C.c_int32(-1)
    rc = L.itts_gemm_tile_occupancy(prec, C.byref(n))
    print(name, "rc", rc, "blocks per CU", n.value)
p = torch.cuda.get_device_properties(0)
print("LDS per block max", getattr(p, "shared_memory_per_block", None), "per multiprocessor", getattr(p, "shared_memory_per_multiprocessor", None), "regs per SM", getattr(p, "regs_per_multiprocessor", None))
