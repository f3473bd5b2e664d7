"""Per-phase shader-clock shares of msda_bwd_mm_kernel (library built with -DDSKD_MM_PROFILE)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tools/prof")
import torch
import msda_only  # noqa: F401  (runs 5 forward + backward launches)
from dskd_amd import native
lib = native.load()
buf = (ctypes.c_ulonglong * 32)()
lib.dskd_debug_mm_prof(buf, 1)
native.msda_backward_raw(*msda_only.args, msda_only.g)
torch.cuda.synchronize()
lib.dskd_debug_mm_prof(buf, 1)
names = ["prologue", "P1 rest", "barrier", "s:data wait | p:products", "s:weights+atomics", "s:fetch issue", "(loop exit)", "flush"]
for role, off in (("wave 0 (sampler)", 0), ("wave 4 (products)", 16)):
    tot = sum(buf[off + i] for i in range(8))
    print(role, "total Mcycles", tot / 1e6)
    for i, n in enumerate(names):
        print(f"   {n:12s} {buf[off + i] / 1e6:10.2f} Mcyc  {100.0 * buf[off + i] / max(tot, 1):5.1f} %")
print("fallback samples: out of window", buf[30], " bad weight quad", buf[31])
