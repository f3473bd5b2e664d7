"""The three fused FFN launches alone (for rocprofv3 --pmc): eval forward, training forward (p=0.1), backward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dskd_amd import native
dev = "cuda"
torch.manual_seed(0)
T = int(os.environ.get("T", 88892))
x = torch.randn(T, 256, device=dev).bfloat16()
w1 = (torch.randn(1024, 256, device=dev) / 16).bfloat16(); b1 = (torch.randn(1024, device=dev) * 0.1).bfloat16()
w2 = (torch.randn(256, 1024, device=dev) / 32).bfloat16(); b2 = (torch.randn(256, device=dev) * 0.1).bfloat16()
gy = torch.randn(T, 256, device=dev).bfloat16()
pf, pb = native.ffn_pack(w1, w2)
for _ in range(int(os.environ.get("N", 5))):
    native.ffn_fwd_raw(x, pf, b1, b2, 0.0, False)
    y, h = native.ffn_fwd_raw(x, pf, b1, b2, 0.1, True)
    native.ffn_bwd_raw(gy, h, pb, 0.1)
torch.cuda.synchronize()
import time
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
if os.environ.get("TIME"):
    print("us eval %.1f  train %.1f  train-p0 %.1f  bwd %.1f" % (
        timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, 0.0, False)), timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, 0.1, True)),
        timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, 0.0, True)), timeit(lambda: native.ffn_bwd_raw(gy, h, pb, 0.1))))
